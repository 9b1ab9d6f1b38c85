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


def param_version(p):
    """Cache key of a parameter's VALUE: (data_ptr, torch version counter) plus the optimizer epoch -- but only for parameters that
    live in a flat optimizer buffer, the only ones updated behind torch's back.  Parameters no optimizer touches (the frozen
    prompter: 231 M values, 12 merged temporal projections) keep their operand copies across steps instead of being re-cast."""
    a = p.data_ptr()
    in_flat = _FLAT_LP["base"] <= a < _FLAT_LP["end"]
    return (param_epoch() if in_flat else -1, a, p._version)


# One flat 16-bit copy of every parameter an optimizer with flat storage owns (alpro_amd.optim.FlatAdamW): refreshed by ONE cast
# launch right after the update instead of ~250 per-tensor casts at first use; operands are then views into it.
_FLAT_LP = {"base": 0, "end": 0, "lp": None, "epoch": -1, "versions": {}}


def register_flat_lp(flat_p, flat_lp, params):
    """flat_lp (16-bit) mirrors flat_p (fp32) element for element as of the current parameter epoch."""
    _FLAT_LP.update(base=flat_p.data_ptr(), end=flat_p.data_ptr() + flat_p.numel() * 4, lp=flat_lp, epoch=param_epoch(),
                    versions={id(p): p._version for p in params})


def _flat_lp_view(plist, dtype):
    """View of the registered flat 16-bit copy covering `plist` (parameters stored back to back), or None."""
    f = _FLAT_LP
    if f["lp"] is None or f["epoch"] != param_epoch() or f["lp"].dtype != dtype:
        return None
    off = None
    nxt = None
    for p in plist:
        a = p.data_ptr()
        if not (f["base"] <= a < f["end"]) or f["versions"].get(id(p)) != p._version or not p.is_contiguous():
            return None
        if nxt is not None and a != nxt:
            return None
        if off is None:
            off = (a - f["base"]) // 4
            if off % 8 != 0:  # the 16-bit view must stay 16-byte aligned for the GEMM operand loads
                return None
        nxt = a + p.numel() * 4
    total = (nxt - f["base"]) // 4 - off
    cols = plist[0].numel() // plist[0].shape[0]
    if any(p.numel() // p.shape[0] != cols for p in plist):
        return None
    return f["lp"][off:off + total].view(-1, cols)


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
        if dtype != torch.float32 and all(p.dim() >= 2 for p in plist):
            v = _flat_lp_view(plist, dtype)
            if v is not None:
                return v
        ver = tuple(param_version(p) for p in plist)
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
