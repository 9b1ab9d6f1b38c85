"""CPU (no GPU): host-side logic -- module tree / state_dict ABI, C-ABI symbol export, loud failure
without a device, row-map arithmetic, the distributed facade."""
import ctypes
import json
import os
import re

import pytest
import torch

from tests.conftest import GOLDEN, ROOT


class Cfg:
    def __init__(self, d):
        self.__dict__.update(d)


VENC = {"cls": "TimeSformer", "patch_size": 16, "attn_drop_rate": 0, "drop_rate": 0, "drop_path_rate": 0.1,
        "maxpool_kernel_size": 2, "use_maxpooling": False, "gradient_checkpointing": False, "img_size": 224}


def make_cfg(bert_cfg, **kw):
    c = Cfg(bert_cfg)
    c.num_entities = 1000
    c.max_n_example_per_group = 1
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_library_exports_every_declared_symbol():
    from alpro_amd import hip
    hdr = open(os.path.join(ROOT, "include", "alpro_hip.h")).read()
    declared = set(re.findall(r"\b(alpro_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().alpro_hip_abi_version() == 4


def test_gemm_desc_matches_header_layout():
    """ctypes struct must mirror alpro_gemm_desc_t field for field."""
    from alpro_amd import hip
    hdr = open(os.path.join(ROOT, "include", "alpro_hip.h")).read()
    body = hdr[hdr.index("typedef struct {"):hdr.index("} alpro_gemm_desc_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(part.replace("*", " ").split()[-1])
    assert names == [f[0] for f in hip.GemmDesc._fields_]
    assert ctypes.sizeof(hip.GemmDesc) == 176


def test_ops_refuse_cpu_tensors():
    from alpro_amd import hip
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.gemm(torch.zeros(4, 768), torch.zeros(4, 768))


def test_state_dict_abi_matches_reference(bert_cfg):
    from alpro_amd.modeling.alpro_models import AlproForPretrain, AlproForVideoTextRetrieval
    keys = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    m = AlproForVideoTextRetrieval(make_cfg(bert_cfg), dict(VENC, num_frm=2))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys["retrieval_T2"]
    sd = m.state_dict()
    assert sd["text_encoder.cls.predictions.decoder.weight"].data_ptr() == sd["text_encoder.bert.embeddings.word_embeddings.weight"].data_ptr()
    assert sd["text_encoder.cls.predictions.decoder.bias"].data_ptr() == sd["text_encoder.cls.predictions.bias"].data_ptr()
    # reference init quirks: temporal_fc zero for blocks > 0 (vit.py:290-298), LN ones/zeros
    blocks = m.visual_encoder.model.blocks
    assert float(blocks[0].temporal_fc.weight.abs().sum()) > 0 and float(blocks[3].temporal_fc.weight.abs().sum()) == 0
    del m
    small = dict(bert_cfg, num_hidden_layers=12)
    m = AlproForPretrain(make_cfg(small), dict(VENC, num_frm=8))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys["pretrain_T8"]
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 465670018


def test_dropout_seed_stream_is_deterministic_and_nonzero():
    from alpro_amd import config as rt
    rt.seed_dropout(42)
    a = [rt.next_dropout_seed() for _ in range(1000)]
    rt.seed_dropout(42)
    assert a == [rt.next_dropout_seed() for _ in range(1000)]
    assert all(0 < x < 2 ** 31 for x in a) and len(set(a)) > 990


def test_dist_single_process_identity():
    from alpro_amd import dist
    x = torch.randn(3, 4, requires_grad=True)
    assert dist.size() == 1 and dist.rank() == 0 and dist.local_rank() == 0
    assert dist.allgather(x) is x
    assert dist.allreduce_grads_([x]) == 0
