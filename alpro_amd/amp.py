"""Loss scaling for the fp16-operand mode of the MI355X path, behind apex.amp's call surface.

Why it exists: BASELINE.json's north star asks for VTC logits within 1e-3 of the fp32 reference.  bf16 operands (8 mantissa bits through
24 GEMM layers) measure 8e-3; fp16 operands (11 bits, same MFMA rate) measure 2.4e-4 -- but fp16's exponent range ends at 6e-5 / 65504,
and the activation gradients of a B = 64 step sit around 1e-3 ... 1e-8.  The reference's own answer for fp16 is apex.amp's dynamic loss
scaling (`with amp.scale_loss(loss, optimizer) as scaled_loss: scaled_loss.backward()`, run_pretrain_sparse.py:596-599, enabled by
`fp16: 1`; apex is not vendored under /root/reference: its documented behaviour is restated here) -- the drivers already call it, so
this module is where the MI355X path plugs in:

  * the loss is multiplied by S (a DEVICE scalar) before backward, so every 16-bit gradient operand (dY of every GEMM and attention
    kernel) carries S * dL/dy; parameter gradients accumulate S * dL/dw in fp32;
  * FlatAdamW.step() folds 1/S into the AdamW kernel's gradient coefficient, detects overflow from the squared gradient norm it
    computes anyway (non-finite -> the update is skipped on the device), and alpro_loss_scale_update moves S (halve on overflow,
    double after 2000 clean steps) -- no host synchronisation anywhere;
  * `scale_loss(..., delay_unscale=False)` (apex's default) leaves TRUE-scale gradients behind at context exit, because the reference's
    drivers clip them themselves before optimizer.step() (run_pretrain_sparse.py:633).

With bf16 / fp32 operands none of this is needed and every function is the identity apex performs when disabled.
"""
import contextlib

import torch

from alpro_amd import config as rt


class LossScaler:
    """Device-resident state {S, growth tracker, applied steps, skipped steps} + the schedule constants (apex defaults)."""

    def __init__(self, init_scale=2.0 ** 16, dynamic=True, growth=2.0, backoff=0.5, window=2000, min_scale=1.0, max_scale=2.0 ** 24, device=None):
        self.dynamic, self.growth, self.backoff, self.window = dynamic, growth, backoff, window
        self.min_scale, self.max_scale = min_scale, max_scale
        self._init = float(init_scale)
        self.state = None
        if device is not None:
            self.to(device)

    def to(self, device):
        if self.state is None or self.state.device != torch.device(device):
            prev = self.state.tolist() if self.state is not None else (getattr(self, "_pending", None) or [self._init, 0.0, 0.0, 0.0])
            self.state = torch.tensor(prev, dtype=torch.float32, device=device)
        return self

    @property
    def scale(self):
        """(1,) device view of S: multiply the loss by it (no host sync)."""
        return self.state[0:1]

    def loss_scale(self):
        return float(self.state[0].item())       # host sync: logging / checkpoints only

    def state_dict(self):
        s = self.state.tolist() if self.state is not None else (getattr(self, "_pending", None) or [self._init, 0.0, 0.0, 0.0])
        return {"loss_scale": s[0], "unskipped": int(s[1]), "applied_steps": int(s[2]), "skipped_steps": int(s[3])}

    def load_state_dict(self, sd):
        vals = [float(sd["loss_scale"]), float(sd.get("unskipped", 0)), float(sd.get("applied_steps", 0)), float(sd.get("skipped_steps", 0))]
        if self.state is None:
            self._init = vals[0]
            self._pending = vals
        else:
            self.state.copy_(torch.tensor(vals, dtype=torch.float32))


_SCALERS = []  # every scaler handed out, in creation order (amp.state_dict() follows apex: one entry per loss scaler)


def needs_loss_scaling(dtype=None):
    return (dtype or rt.compute_dtype()) == torch.float16


def scaler_for(optimizer, create=True, **kw):
    """The LossScaler attached to `optimizer` (through the hvd.DistributedOptimizer facade if present)."""
    inner = getattr(optimizer, "_opt", optimizer)
    sc = getattr(inner, "scaler", None)
    if sc is None and create:
        sc = LossScaler(**kw)
        inner.scaler = sc
        _SCALERS.append(sc)
        rt.set_armed_loss_scaler(sc)
    return sc


def initialize(models, optimizers=None, enabled=True, opt_level="O1", loss_scale="dynamic", **unused):
    """apex.amp.initialize.  The precision policy of this path is ALPRO_COMPUTE_DTYPE, not opt_level: with fp16 operands every optimizer
    gets a LossScaler (dynamic, or the fixed value given as loss_scale); otherwise nothing changes.  Returns what apex returns."""
    if optimizers is not None and needs_loss_scaling():
        for opt in (optimizers if isinstance(optimizers, (list, tuple)) else [optimizers]):
            if loss_scale == "dynamic" or loss_scale is None:
                scaler_for(opt)
            else:
                scaler_for(opt, init_scale=float(loss_scale), dynamic=False)
    return models if optimizers is None else (models, optimizers)


def _grads_of(optimizer):
    from alpro_amd.optim import real_grad
    return [p.grad for g in optimizer.param_groups for p in g["params"] if real_grad(p) is not None]


def unscale_(optimizer, scaler):
    """Divide the gradients by S in place (apex does this at scale_loss exit).  FlatAdamW: one elementwise pass over the flat buffer and a
    note to step() that the 1/S is already applied; any other optimizer: a foreach multiply over its gradients."""
    inner = getattr(optimizer, "_opt", optimizer)
    inv = torch.reciprocal(scaler.scale).reshape(())
    flat = getattr(inner, "flat", None)
    if flat is not None:
        if getattr(inner, "_inflight", None):
            # the overlapped exchange launched from inside backward (FlatAdamW._on_grads_final) is still writing this buffer on RCCL's stream:
            # finish it (remaining ranges go out, every handle is waited for, a 16-bit wire copy comes back) BEFORE scaling in place, and tell
            # step() that the summed gradients are already here -- a driver that calls optimizer.synchronize() inside the with block (the
            # reference's do) never gets here with handles pending
            inner._finish_exchange()
            inner._note_synced("sum")
        flat["g"].mul_(inv)
    else:
        grads = _grads_of(inner)
        if grads:
            torch._foreach_mul_(grads, inv)
    inner._grads_scaled = False


@contextlib.contextmanager
def scale_loss(loss, optimizers, delay_unscale=False, **unused):
    """`with amp.scale_loss(loss, optimizer) as scaled_loss: scaled_loss.backward()`.  Identity unless the operands are fp16."""
    if not needs_loss_scaling():
        yield loss
        return
    opts = optimizers if isinstance(optimizers, (list, tuple)) else [optimizers]
    scaler = scaler_for(opts[0]).to(loss.device)
    for o in opts[1:]:
        getattr(o, "_opt", o).scaler = scaler
    for o in opts:
        getattr(o, "_opt", o)._grads_scaled = True
        if hasattr(getattr(o, "_opt", o), "_g_clean"):
            getattr(o, "_opt", o)._mark_dirty()      # gradients are about to be written by the caller's own scaled_loss.backward()
    with rt.loss_scaling(scaler):
        yield loss * scaler.scale.reshape(())
    if not delay_unscale:
        for o in opts:
            unscale_(o, scaler)
            _install_foreign_step_guard(getattr(o, "_opt", o), scaler)


def _install_foreign_step_guard(opt, scaler):
    """Optimizers other than FlatAdamW (the reference's own AdamW through the launcher) know nothing about overflow: wrap step() once so
    that a step whose gradients are not finite is skipped and the scale schedule advances -- apex patches optimizer.step the same way.
    Costs one host sync per step on that (non-fused) path."""
    if hasattr(opt, "flat") or getattr(opt, "_alpro_guarded", False):
        return
    real = opt.step

    def step(*a, **k):
        grads = _grads_of(opt)
        norms = torch._foreach_norm(grads) if grads else []
        finite = bool(torch.isfinite(torch.stack(norms)).all()) if norms else True
        st = scaler.state
        if not finite:
            st[0] = max(float(st[0]) * scaler.backoff, scaler.min_scale) if scaler.dynamic else st[0]
            st[1] = 0.0
            st[3] += 1.0
            return None
        out = real(*a, **k)
        st[2] += 1.0
        st[1] += 1.0
        if scaler.dynamic and float(st[1]) >= scaler.window:
            st[0] = min(float(st[0]) * scaler.growth, scaler.max_scale)
            st[1] = 0.0
        return out
    opt.step = step
    opt._alpro_guarded = True


def master_params(optimizer):
    """apex.amp.master_params: what the drivers hand to clip_grad_norm_ (run_pretrain_sparse.py:633).  FlatAdamW answers with ONE view over
    its flat gradient buffer; placeholder gradients (alpro_amd.optim.zero_none_grad) are left out -- they are zeros and must not be scaled
    in place (every element aliases one scalar)."""
    from alpro_amd.optim import is_placeholder_grad
    inner = getattr(optimizer, "_opt", optimizer)
    if hasattr(inner, "master_params"):
        yield from inner.master_params()
        return
    for group in optimizer.param_groups:
        for p in group["params"]:
            if not is_placeholder_grad(p.grad):
                yield p


def state_dict():
    return {"loss_scaler%d" % i: sc.state_dict() for i, sc in enumerate(_SCALERS)}


def load_state_dict(sd):
    for i, sc in enumerate(_SCALERS):
        if sd and ("loss_scaler%d" % i) in sd:
            sc.load_state_dict(sd["loss_scaler%d" % i])
