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


# Parameters updated by anything ELSE than FlatAdamW.  The reference's own optimizer (src/optimization/adamw.py:88,101) writes through
# `p.data.addcdiv_` / `p.data.add_`: `.data` carries its own version counter, so p._version does not move and a cache keyed on it
# alone would hand the forward the INITIAL weights forever.  Every torch.optim.Optimizer.step() (any subclass, the reference's AdamW
# included) therefore bumps this second epoch through a global post-step hook; the hvd.DistributedOptimizer facade and drivers with
# hand-rolled updates call notify_params_updated() themselves.
_EXT_EPOCH = [0]


def notify_params_updated():
    """Parameters outside FlatAdamW's flat buffer may have changed in place: their operand copies are re-cast at next use."""
    _EXT_EPOCH[0] += 1


def _install_optimizer_hook():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(lambda opt, args, kwargs: notify_params_updated())
    except Exception as e:  # noqa: BLE001 -- very old torch: the facade / the driver must call notify_params_updated()
        import warnings
        warnings.warn("alpro_amd: torch.optim's global post-step hook is unavailable (%r): after every in-place parameter update outside "
                      "alpro_amd.optim.FlatAdamW call alpro_amd.modeling.weights.notify_params_updated(), or the 16-bit operand copies of the "
                      "weights go stale without an error (INTEGRATION.md, 'Manual parameter updates')" % (e,))


_install_optimizer_hook()


def param_version(p):
    """Cache key of a parameter's VALUE: (epoch tag, data_ptr, torch version counter).  Parameters inside FlatAdamW's flat buffer are
    updated through raw pointers: tag = the flat optimizer epoch (>= 0).  All others: tag = -1 - (external epoch), which only moves when
    some torch optimizer stepped (see above) -- so under FlatAdamW the parameters no optimizer touches (the frozen prompter: 231 M
    values, 12 merged temporal projections) keep their operand copies across steps, and under the reference's AdamW every copy is
    refreshed after every step.  Parameters that do not require gradients never go stale through an optimizer: tag -1."""
    a = p.data_ptr()
    if _FLAT_LP["base"] <= a < _FLAT_LP["end"]:
        return (param_epoch(), a, p._version)
    return (-1 - _EXT_EPOCH[0] if p.requires_grad else -1, a, p._version)


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


# Counts operand copies (re)built since import: the two-stream block runner (modeling/timesformer/vit.py::run_blocks) orders its second stream
# behind the launch stream whenever a block built one (the copy kernels run on the stream that missed).
_REBUILDS = [0]


def operand_rebuilds():
    return _REBUILDS[0]


def note_operand_rebuild():
    _REBUILDS[0] += 1


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
        note_operand_rebuild()
        return out

    def clear(self):
        self._store.clear()
