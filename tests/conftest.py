import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-minute CPU test")


GOLDEN = os.path.join(ROOT, "tests", "golden")

BERT_CFG = {  # config_release/base_model.json (reference) -- the JSON surface is kept verbatim
    "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu", "hidden_dropout_prob": 0.1, "hidden_size": 768,
    "initializer_range": 0.02, "intermediate_size": 3072, "layer_norm_eps": 1e-12, "max_position_embeddings": 512,
    "model_type": "bert", "num_attention_heads": 12, "num_hidden_layers": 12, "pad_token_id": 0,
    "type_vocab_size": 2, "vocab_size": 30522, "fusion_layer": 6, "encoder_width": 768, "itc_token_type": "cls",
}


@pytest.fixture(scope="session")
def bert_cfg():
    return dict(BERT_CFG)
