"""Operand copies of fp32 master parameters in the compute dtype.

Parameters stay fp32 nn.Parameters under the reference's state_dict names.  The MFMA operands
are 16-bit copies refreshed by alpro_cast_from_f32 whenever a parameter's in-place version
counter changes (i.e. after every optimizer step); in exact (fp32) mode the masters are used
directly.  Several parameters can be fused into one operand (BERT q/k/v -> one (3H, H) weight).
"""
import torch

from alpro_amd import hip


# Bumped by optimizers that update parameters through raw pointers (alpro_amd.optim.FlatAdamW): such updates do
# not touch torch's per-tensor version counters, so the operand copies are keyed on this epoch as well.
_PARAM_EPOCH = [0]


def bump_param_epoch():
    _PARAM_EPOCH[0] += 1


def param_epoch():
    return _PARAM_EPOCH[0]


class OperandCache:
    def __init__(self):
        self._store = {}

    def get(self, key, params, dtype):
        """params: tensor or tuple of tensors concatenated along dim 0; returns a `dtype` device tensor."""
        single = torch.is_tensor(params)
        plist = (params,) if single else tuple(params)
        if single and dtype == torch.float32:
            w = params.detach()
            return w if w.is_contiguous() else w.contiguous()
        ver = (param_epoch(),) + tuple((p.data_ptr(), p._version) for p in plist)
        hit = self._store.get(key)
        if hit is not None and hit[0] == ver and hit[1].dtype == dtype:
            return hit[1]
        with torch.no_grad():
            src = plist[0].detach() if single else torch.cat([p.detach().reshape(p.shape[0], -1) if p.dim() > 1 else p.detach() for p in plist], 0)
            src = src.contiguous().float()
            out = hip.cast(src, dtype) if dtype != torch.float32 else src
        self._store[key] = (ver, out)
        return out

    def clear(self):
        self._store.clear()
