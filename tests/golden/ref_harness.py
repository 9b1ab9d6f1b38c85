"""Import the read-only reference (/root/reference) on CPU in the AUTHORING container only.

Used by make_golden.py to capture golden vectors. This file never travels to the GPU box in
a usable form (it needs /root/reference) and nothing under alpro_amd/, bench.py or the -m gpu
tests imports it. The shims below only replace packages that are absent from this image
(horovod, apex, easydict, ujson, tensorboardX) or API names that transformers 5.x dropped;
none of them changes arithmetic.
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF = os.environ.get("ALPRO_REFERENCE", "/root/reference")

_state = {"rank": 0, "local_rank": 0, "size": 1, "gathered": None}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _allgather(x, name=None):
    """Single-process stand-in for hvd.allgather: identity at size 1; for simulated W ranks the
    caller pre-registers the other ranks' features (set_sim_ranks)."""
    g = _state["gathered"]
    if g is None:
        return x
    lst = g["video"] if g["turn"] % 2 == 0 else g["text"]  # alpro_models.py:110-111 order
    g["turn"] += 1
    out = [x if i == _state["local_rank"] else t for i, t in enumerate(lst)]
    return torch.cat(out, 0)


def install_shims():
    import transformers  # noqa: F401  (must precede the tensorboardX stub; accelerate probes it)
    hvd_t = _mod("horovod.torch", allgather=_allgather, rank=lambda: _state["rank"],
                 local_rank=lambda: _state["local_rank"], size=lambda: _state["size"],
                 init=lambda: None)
    _mod("horovod.torch.mpi_ops")
    _mod("horovod", torch=hvd_t)
    _mod("apex.amp")
    fln = _mod("apex.normalization.fused_layer_norm", FusedLayerNorm=nn.LayerNorm)
    norm = _mod("apex.normalization", fused_layer_norm=fln)
    _mod("apex", normalization=norm, amp=sys.modules["apex.amp"])
    _mod("ujson", load=json.load, loads=json.loads, dump=json.dump, dumps=json.dumps)

    class _SW:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return lambda *a, **k: None

    _mod("tensorboardX", SummaryWriter=_SW)

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in {**(d or {}), **kw}.items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __getattr__ = dict.__getitem__
        __setattr__ = __setitem__

    _mod("easydict", EasyDict=EasyDict)

    import transformers.file_utils as fu
    import transformers.modeling_utils as mu
    from transformers import pytorch_utils as pu

    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    mu.PreTrainedModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n

    def _noop_deco(*a, **k):
        def deco(fn):
            return fn
        return deco

    for n in ("add_code_sample_docstrings", "add_start_docstrings",
              "add_start_docstrings_to_model_forward", "replace_return_docstrings"):
        setattr(fu, n, _noop_deco)
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    if REF not in sys.path:
        sys.path.insert(0, REF)


def import_reference():
    install_shims()
    import src.modeling.xbert as xbert

    def _init_weights_v4(self):
        self.apply(self._init_weights)

    xbert.BertPreTrainedModel.init_weights = _init_weights_v4
    xbert.BertPreTrainedModel.post_init = lambda self: None

    _orig_init = xbert.BertForMaskedLM.__init__

    def _mlm_init(self, config):
        _orig_init(self, config)
        # v4 semantics: tie decoder to word embeddings (xbert.py:670-677 + PreTrainedModel.tie_weights)
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    xbert.BertForMaskedLM.__init__ = _mlm_init
    xbert.BertForMaskedLM.get_input_embeddings = lambda self: self.bert.embeddings.word_embeddings
    xbert.BertForMaskedLM.from_pretrained = classmethod(lambda cls, name, config=None, **kw: cls(config))
    xbert.BertModel.from_pretrained = classmethod(lambda cls, name, config=None, **kw: cls(config, **kw))
    import src.modeling.alpro_models as am
    return am, xbert


def make_configs(num_frm=8, img_size=224, num_entities=1000, **bert_overrides):
    from transformers import BertConfig
    base = json.load(open(os.path.join(REF, "config_release/base_model.json")))
    base.update(bert_overrides)
    cfg = BertConfig(**base)
    cfg.num_entities = num_entities
    cfg.max_n_example_per_group = 1
    venc = json.load(open(os.path.join(REF, "config_release/timesformer_divst_8x32_224_k600.json")))
    venc.update(num_frm=num_frm, img_size=img_size)
    return cfg, venc


def set_sim_ranks(local_rank, video_feats, text_feats):
    """Simulate W ranks on one process: lists of per-rank (B,256) features."""
    _state["local_rank"] = local_rank
    _state["rank"] = local_rank
    _state["size"] = len(video_feats)
    _state["gathered"] = {"video": video_feats, "text": text_feats, "turn": 0}


def clear_sim_ranks():
    _state.update(rank=0, local_rank=0, size=1, gathered=None)
