"""Training-side plumbing shared by the ViT and BERT blocks: gradient accumulation into `.grad`,
Linear backward on the NT GEMM (dgrad with a cached transposed operand, wgrad on transposed
activations with the bias gradient fused into the transpose), and the autograd anchor.

Why not torch.autograd per op: the fused epilogues write through row maps into shared token buffers,
so each block gets ONE hand-written backward built from the C-ABI kernels (alpro_attn_bwd,
alpro_layernorm_bwd, alpro_gemm, alpro_transpose, ...).  Parameter gradients are accumulated in place
into `param.grad` (fp32) by the wgrad GEMM's residual input, which is what `loss.backward()` leaves
behind in the reference (run_pretrain_sparse.py:599), ready for the all-reduce and the optimizer.
"""
import os
import weakref

import torch

from alpro_amd import hip
from alpro_amd.modeling.weights import param_version


def grad_buffer(p, zero=False):
    """Return (p.grad, existed).  Allocates an fp32 buffer on first use."""
    if p.grad is not None and (p.grad.numel() <= 1 or any(p.grad.stride())):   # (a stride-0 placeholder of optim.zero_none_grad is "no buffer yet")
        return p.grad, True
    p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format) if zero else torch.empty_like(p, memory_format=torch.contiguous_format)
    return p.grad, False


def add_grad(p, g):
    if p.grad is None or (p.grad.numel() > 1 and not any(p.grad.stride())):
        p.grad = g.detach().clone().reshape(p.shape).contiguous()
    else:
        p.grad.add_(g.reshape(p.shape))


def fused_param_view(params):
    """If `params` (equal trailing shape, contiguous) lie back to back in one storage (FlatAdamW's flat parameter buffer), return a
    single (sum of rows, cols) view over their values, else None."""
    ps = [p.detach() for p in params]
    if any(not p.is_contiguous() for p in ps):
        return None
    base = ps[0]
    off = base.data_ptr()
    for p in ps:
        if p.untyped_storage().data_ptr() != base.untyped_storage().data_ptr() or p.data_ptr() != off or p.shape[1:] != base.shape[1:]:
            return None
        off += p.numel() * 4
    rows = sum(p.shape[0] for p in ps)
    return torch.as_strided(base, (rows, base.numel() // base.shape[0]), (base.numel() // base.shape[0], 1))


# W^T operands that live across steps: (cache, key) -> entry.  After an optimizer step ALL of them are stale at once, so
# refresh_transposed_operands() rewrites them with ONE alpro_transpose_batch launch instead of ~130 launches at first use.
_WT_REGISTRY = {}
_WT_TABLE = {}  # out dtype -> dict(sig, table, njobs, tiles): the device job table, rebuilt only when the set of operands changes


def transposed_operand(cache, key, weight, dt):
    """(N, K) fp32 parameter -- or a tuple of parameters concatenated along dim 0 -- -> cached (K, N64) operand in `dt` for dgrad
    (dX = dY @ W), refreshed when the parameter value changes (weights.param_version)."""
    plist = (weight,) if torch.is_tensor(weight) else tuple(weight)
    ver = tuple(param_version(p) for p in plist)
    hit = cache._store.get(key)
    if hit is not None and hit[0] == ver and hit[1].dtype == dt:
        return hit[1]
    with torch.no_grad():
        src = fused_param_view(plist) if len(plist) > 1 else plist[0].detach().reshape(plist[0].shape[0], -1)
        if src is None:
            src = torch.cat([p.detach().reshape(p.shape[0], -1) for p in plist], 0)
        src = src.contiguous()
        out = hip.transpose(src, out_dtype=dt, pad_to=64)
    cache._store[key] = (ver, out)
    # a view of the parameter storage itself, of parameters an optimizer with flat storage updates (the only ones that go stale
    # every step; frozen ones never do): re-read by the batched refresh
    stable = src.data_ptr() == plist[0].data_ptr() and all(v[0] >= 0 for v in ver)
    if stable and src.dtype == torch.float32 and src.is_cuda:
        _WT_REGISTRY[(id(cache), key)] = dict(cache=weakref.ref(cache), key=key, params=[weakref.ref(p) for p in plist], src=src, out=out)
    else:
        _WT_REGISTRY.pop((id(cache), key), None)
    return out


def refresh_transposed_operands():
    """Called by the optimizer right after it updated the parameters: rewrite every registered W^T operand in place with one
    batched launch and stamp it with the new parameter versions.  Entries whose module is gone, whose parameters moved, or whose
    operand was replaced meanwhile are dropped (they fall back to the lazy path above)."""
    if os.environ.get("ALPRO_WT_REFRESH", "1") == "0":  # measurement knob: back to one transpose launch per Linear at first use
        return 0
    live = []
    for k, e in list(_WT_REGISTRY.items()):
        cache, ps = e["cache"](), [r() for r in e["params"]]
        hit = cache._store.get(e["key"]) if cache is not None else None
        if hit is None or hit[1] is not e["out"] or any(p is None for p in ps) or ps[0].data_ptr() != e["src"].data_ptr():
            del _WT_REGISTRY[k]
            continue
        live.append((e, cache, ps))
    if not live:
        return 0
    t = _WT_TABLE
    by_dtype = {}
    for item in live:
        by_dtype.setdefault(item[0]["out"].dtype, []).append(item)
    n = 0
    for dt, items in by_dtype.items():
        sig = tuple((e["src"].data_ptr(), e["out"].data_ptr()) for e, _, _ in items)
        slot = t.setdefault(dt, {})
        if slot.get("sig") != sig:
            slot["table"], slot["njobs"], slot["tiles"] = hip.transpose_jobs([(e["src"], e["out"]) for e, _, _ in items])
            slot["sig"] = sig
        hip.transpose_batch(slot["table"], slot["njobs"], slot["tiles"], dt)
        for e, cache, ps in items:
            cache._store[e["key"]] = (tuple(param_version(p) for p in ps), e["out"])
        n += len(items)
    return n


def bias_grad(bias):
    """fp32 .grad buffer of a bias (zero-initialised on first use): the `colsum` target of the kernel that PRODUCES the
    Linear's output gradient (alpro_gather_cast / alpro_gemm epilogue), so dY is not read a second time for it."""
    return grad_buffer(bias, zero=True)[0]


def fused_grad_view(params):
    """If the fp32 .grad buffers of `params` (equal trailing shape) lie back to back in one storage (FlatAdamW's flat
    gradient buffer), return a single (sum of rows, ...) view over them, else None."""
    gs = [p.grad for p in params]
    if any(g is None or not g.is_contiguous() for g in gs):
        return None
    base = gs[0]
    off = base.data_ptr()
    for g in gs:
        if g.untyped_storage().data_ptr() != base.untyped_storage().data_ptr() or g.data_ptr() != off or g.shape[1:] != base.shape[1:]:
            return None
        off += g.numel() * 4
    rows = sum(g.shape[0] for g in gs)
    return torch.as_strided(base, (rows,) + tuple(base.shape[1:]), base.stride())


def wgrad(dy_t, x_t, weight, bias, bias_done=False):
    """weight.grad += dy^T x ; bias.grad += colsum(dy) unless the producer of dy already did (bias_done).
    dy_t (M, N), x_t (M, K) operand-dtype activations.  16-bit operands: alpro_gemm_tn_acc reads both in place (split
    over tokens, fp32 atomics) and takes the bias gradient from the dY fragments it holds; fp32 (exact mode): transposed
    copies + the NT GEMM, bias gradient fused into the transpose."""
    if bias_done:
        bias = None
    if bias is not None:
        gb = grad_buffer(bias, zero=True)[0]
    gw, existed = grad_buffer(weight, zero=True)
    gw2 = gw.view(gw.shape[0], -1)
    if dy_t.dtype != torch.float32:
        from alpro_amd import config as rt
        side = rt.wgrad_side_stream(dy_t.device)
        if side is None:
            hip.gemm_tn_acc(dy_t, x_t, gw2, colsum=gb if bias is not None else None)  # bias gradient from the same pass over dy
            return
        # side stream (alpro_amd.config, ALPRO_WGRAD_STREAM): behind everything the launch stream has queued so far (dy, x, the zeroed buffers)
        side.wait_stream(torch.cuda.current_stream(dy_t.device))
        with torch.cuda.stream(side):
            hip.gemm_tn_acc(dy_t, x_t, gw2, colsum=gb if bias is not None else None)
        dy_t.record_stream(side)   # the launch stream's allocator may not hand these blocks out again before the side stream is done with them
        x_t.record_stream(side)
        return
    dyT = hip.transpose(dy_t, colsum=gb if bias is not None else None)
    hip.gemm(dyT, hip.transpose(x_t), out=gw2, out_dtype=torch.float32, residual=gw2)


SAVE_GELU_GRAD = os.environ.get("ALPRO_SAVE_GELU_GRAD", "1") != "0"   # GELU Linears keep gelu'(pre-activation) instead of the pre-activation (0 = round-2 form, A/B)


TILED_GELU_GRAD = os.environ.get("ALPRO_TILED_GELU_GRAD", "1") != "0"   # the saved gelu' lives in the GEMM's tile layout where the library offers it (0 = rows, A/B)


def gelu_save_buffer(M, N, K, dt, device):
    """(buffer, tiled) for what a GELU Linear (M, K) -> (M, N) keeps for its backward.  With SAVE_GELU_GRAD the buffer holds gelu'(..) and only
    the two GEMM epilogues ever look at it (GELU_SAVE_GRAD writes, MUL_SAVED of the following Linear's dgrad reads): where the library runs both
    on its 8-phase kernel it is kept in that kernel's private tile order (hip.gemm c2_tiled) and never crosses the LDS."""
    rows = hip.gemm_c2_tiled_rows(M, N, K, dt) if (SAVE_GELU_GRAD and TILED_GELU_GRAD) else 0
    return torch.empty((rows or M, N), dtype=dt, device=device), rows > 0


def dgrad(dy_t, wT, out_dtype=None, row_scale=None, row_scale_group=1, gelu_pre=None, gelu_saved_grad=False, gelu_tiled=False):
    """dX = dy @ W using the cached transposed operand wT (K, N64); dy_t (M, N).
    gelu_pre: what the forward of the GELU Linear that produced this Linear's input kept in its C2 buffer -- the pre-activation
    (dX *= gelu'(pre) recomputed in the epilogue) or, with gelu_saved_grad, gelu'(pre) itself (dX *= saved, ALPRO_ACT_MUL_SAVED).
    Either way the elementwise GELU backward never runs as its own pass."""
    n = dy_t.shape[1]
    w = wT if wT.shape[1] == n else wT[:, :n]
    if not w.is_contiguous() or n % (64 if dy_t.dtype != torch.float32 else 32) != 0:
        # contraction dim not a K-granule multiple (heads only): pad both operands with zero columns
        pad = wT.shape[1] - n
        dy_t = torch.nn.functional.pad(dy_t, (0, pad))
        w = wT
    return hip.gemm(dy_t, w, out_dtype=out_dtype or dy_t.dtype, row_scale=row_scale, row_scale_group=row_scale_group,
                    act=(hip.ACT_MUL_SAVED if gelu_saved_grad else hip.ACT_GELU_BWD) if gelu_pre is not None else hip.ACT_NONE, pre_act=gelu_pre,
                    c2_tiled=gelu_tiled and gelu_saved_grad)


# Anchored runs whose backward has not executed yet (weak: an abandoned graph drops out when it is freed).  A run may only declare
# gradients OUTSIDE itself final -- FlatAdamW then puts them on the wire -- when no other anchored backward is still to come:
# autograd's node order is not a promise (AlproForSequenceClassification anchors the text encoder BEFORE the visual encoder, so the
# visual node runs first), and a gradient range that is all-reduced before its backward wrote it makes the replicas diverge silently.
_LIVE_ANCHORS = weakref.WeakSet()


BACKWARD_EPOCH = [0]   # bumped by every anchored backward: "parameter gradients may have been written since" (FlatAdamW's fused zero_grad bookkeeping)


class Anchor(torch.autograd.Function):
    """Ties a hand-written encoder backward into torch.autograd.

    forward(run, n_act, *inputs): `run.forward(*activations)` computes the outputs with the HIP kernels (under
    no_grad) and stores what its backward needs on `run`; the parameters ride along only so that autograd
    schedules the node.  backward: `run.backward(*grad_outputs)` returns gradients for the activation inputs
    and accumulates parameter gradients directly into `.grad`.
    """

    @staticmethod
    def forward(ctx, run, n_act, *inputs):
        ctx.run = run
        ctx.n_act = n_act
        ctx.n_in = len(inputs)
        from alpro_amd import config as rt
        ctx.dt = rt.compute_dtype()
        ctx.set_materialize_grads(False)  # an unused output (mlm_scores: 312 MB at B = 64) arrives as None, not as a zero tensor
        _LIVE_ANCHORS.add(ctx)
        with torch.no_grad():
            outs = run.forward(*inputs[:n_act])
        ctx.single = torch.is_tensor(outs)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        # what the run may assume about the rest of the backward pass: True = some other anchored backward has not run yet
        ctx.run.others_pending = any(c is not ctx for c in _LIVE_ANCHORS)
        BACKWARD_EPOCH[0] += 1
        from alpro_amd import config as rt
        rt.check_backward_precision(getattr(ctx.run, "dt", None) or ctx.dt)
        rt.wgrad_scope(True)
        try:
            with torch.no_grad():
                g = ctx.run.backward(*[None if x is None else x.contiguous() for x in grads])
        finally:
            rt.wgrad_scope(False)
            rt.join_wgrad()   # weight gradients launched on the side stream: the launch stream has them from here on
            _LIVE_ANCHORS.discard(ctx)
        if grads and any(x is not None and x.is_cuda for x in grads):
            cur = torch.cuda.current_stream()
            if rt.is_text_side_stream(cur):
                # this backward ran on the text side stream (autograd: the stream of the node's forward) and wrote parameter gradients behind autograd's
                # back: the caller's stream takes them over when the whole backward pass is done (final callbacks run on the caller's streams)
                ev = cur.record_event()
                torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream().wait_event(ev))
        g = (g,) if (torch.is_tensor(g) or g is None) else tuple(g)
        g = g + (None,) * (ctx.n_act - len(g))
        ctx.run = None
        return (None, None) + tuple(g) + (None,) * (ctx.n_in - ctx.n_act)


def run_anchored(run, activations, params):
    """Call `run` through autograd.  activations: tensors that may need gradients; params: nn.Parameters."""
    params = [p for p in params if p.requires_grad]
    need = torch.is_grad_enabled() and (any(torch.is_tensor(a) and a.requires_grad for a in activations) or len(params) > 0)
    if not need:
        with torch.no_grad():
            return run.forward(*activations)
    return Anchor.apply(run, len(activations), *activations, *params)
