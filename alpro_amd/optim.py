"""Step epilogue of ALPRO's data-parallel training loop, MI355X-first.

The reference does, per optimizer step (run_pretrain_sparse.py:595-648): zero_none_grad -> Horovod all-reduce of
every parameter gradient -> LR set -> clip_grad_norm_ -> HF-style AdamW.step (src/optimization/adamw.py:40-103, ~930
tensors x ~8 tiny launches) -> zero_grad.  Here all trainable parameters, their gradients and the Adam moments live in
FOUR flat fp32 buffers (0.94 GB each for AlproForPretrain's 234 M trained parameters; HBM is 288 GB):

  * the wgrad GEMMs accumulate straight into views of the flat gradient buffer,
  * the gradient all-reduce is a handful of large RCCL calls over that buffer (no bucketing copies),
  * grad-norm + clip + AdamW are two kernels (alpro_sumsq, alpro_adamw_step), zero_grad is one memset.

Same constructor / param_groups surface as the reference's AdamW so a driver only swaps the class.  Parameters that never
receive a gradient (the frozen prompter, the unused Kinetics head) are left out instead of being zero-filled and reduced.
"""
import math

import torch

from alpro_amd import dist, hip
from alpro_amd import config as rt
from alpro_amd.modeling.weights import bump_param_epoch, register_flat_lp


# ---- placeholder gradients: what the reference's `zero_none_grad` (src/utils/misc.py:28-31) stands for, without the bytes ------------------
# The reference's drivers fill the gradient of every parameter that received none (the frozen prompter: 231 M values; the unused Kinetics
# head) with zeros before the exchange, and assert afterwards that no trainable parameter has `grad is None` (run_pretrain_sparse.py:598,
# 640-644).  Here such a parameter gets a stride-0 view of ONE shared zero scalar: `grad is not None` holds, its value is the zero tensor of
# the right shape, and everything on this path that would move or update gradients (FlatAdamW's flat buffers and exchange, the hvd facade's
# buckets, amp's unscale, master_params for clipping) skips it -- a zero gradient moves neither the Adam moments nor the parameter
# (adamw.py:77-88 with weight_decay 0), so skipping it is what the reference computes, minus 0.93 GB of zeros on the wire per step.
_ZERO = {}


def placeholder_grad(p):
    """Stride-0 zero gradient for `p` (shared storage per device / dtype); one-element parameters get a real zero (nothing to save)."""
    if p.numel() <= 1:
        return torch.zeros_like(p)
    key = (p.device, p.dtype)
    z = _ZERO.get(key)
    if z is None:
        z = _ZERO[key] = torch.zeros(1, dtype=p.dtype, device=p.device)
    return z.expand(p.shape)


def is_placeholder_grad(g):
    return g is not None and g.dim() > 0 and g.numel() > 1 and not any(g.stride())


def zero_none_grad(model):
    """Drop-in for src.utils.misc.zero_none_grad (reference misc.py:28-31): afterwards no trainable parameter has `grad is None`."""
    for p in model.parameters():
        if p.grad is None and p.requires_grad:
            p.grad = placeholder_grad(p)


def real_grad(p):
    """p.grad unless it is absent or a placeholder."""
    g = p.grad
    return None if (g is None or is_placeholder_grad(g)) else g


def _backward_epoch():
    from alpro_amd.modeling import train as tr
    return tr.BACKWARD_EPOCH[0]


class _FlatView:
    """What amp.master_params() yields for a FlatAdamW whose flat buffers exist: ONE object whose .grad is the whole flat gradient buffer, so
    the driver's `clip_grad_norm_(amp.master_params(optimizer), cfg.grad_norm)` (run_pretrain_sparse.py:633) is one norm and one scale over
    0.94 GB instead of ~460 tensors (torch's clip only reads `.grad`)."""

    def __init__(self, grad):
        self.grad = grad


class FlatAdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, max_grad_norm=None,
                 allreduce=True, bucket_elems=64 << 20, overlap_backward=None, wire_dtype=None, fused_zero_grad=True):
        self.params = [p for p in params if p.requires_grad]
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)]
        self.max_grad_norm = max_grad_norm
        self.allreduce = allreduce
        self.bucket_elems = bucket_elems
        # overlap_backward: slices of the flat gradient buffer are all-reduced (async, RCCL's own stream) as soon as the backward
        # code reports them final (dist.grads_final): BERT + heads before the ViT backward starts, ViT blocks four at a time.
        # Needs ONE backward per step (gradient accumulation over several backward passes: pass overlap_backward=False).
        # wire_dtype=torch.bfloat16 halves the bytes on xGMI (cast -> all-reduce -> cast back; sums of <= 8 ranks in bf16).
        if overlap_backward is None:   # default on; ALPRO_OVERLAP_BACKWARD=0 exchanges after backward instead (see DESIGN.md section 6 on CU contention)
            import os
            overlap_backward = os.environ.get("ALPRO_OVERLAP_BACKWARD", "1") != "0"
        self.overlap_backward = overlap_backward
        self.wire_dtype = wire_dtype
        self._inflight, self._reduced = [], []     # async handles (+ staging tensors) and element ranges already on the wire
        # fused_zero_grad: step() hands the flat gradient buffer back ZEROED (alpro_adamw_step clears it on the way: the gradients are consumed
        # by the update, as the reference's `optimizer.step(); optimizer.zero_grad()` pair leaves them, run_pretrain_sparse.py:646-648), and the
        # zero_grad() that follows is free instead of a second 0.94 GB pass
        self.fused_zero_grad = fused_zero_grad
        self._g_clean = False
        self._clean_epoch = -1
        self._cus_reserved = False
        self.step_count = 0
        self._pending_state = None  # load_state_dict() before the flat buffers exist: applied by _build()
        self.flat = None  # built at the first step(), when we know which parameters actually receive gradients
        self.last_grad_norm = None
        self._pre_synced = None  # "sum" / "avg": synchronize() already ran for this step (hvd-style drivers call it themselves)
        # ... and that only holds while nothing has written a gradient since: the reference's drivers call optimizer.synchronize() after
        # EVERY micro-step of a gradient-accumulation window (run_pretrain_sparse.py:596-601 with gradient_accumulation_steps = 2 in
        # config_release/msrvtt_qa.json, msvd_qa.json, pretrain_prompter.json), and each of those calls has to exchange what the backward
        # before it added.  _writes counts "gradients may have been written" events (autograd's accumulation hooks, the
        # hand-written backward passes' grads_final reports, backward(), amp.scale_loss); the stamp is (_writes, BACKWARD_EPOCH) at sync time.
        self._writes = 0
        self._sync_stamp = None
        # exchange diagnostics (round 6: the N > 1 bench line reads them, tests/test_bench_multirank.py asserts them): per finished exchange, how
        # much went on the wire from inside backward, how much only at synchronize(), the CU budget in effect meanwhile, and how long the
        # compute stream was held by the exchange after backward had returned (device events around the late launches and the handle waits)
        self.record_exchange = False
        self.exchange_log = []
        self._early = [0, 0]           # ranges / elements launched by _on_grads_final since the last finished exchange
        self._budget_seen = None
        if hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            for p in self.params:   # from construction on (not from _build() on): the first step's micro-steps run before the flat buffers exist
                p.register_post_accumulate_grad_hook(self._mark_dirty)
        # fp16 operands: dynamic loss scaling (alpro_amd/amp.py).  `scaler` is attached here when the compute dtype already is fp16 (so
        # that forward-time gradient producers -- the LM head -- see the scale), else by the first backward() / amp.scale_loss().
        self.scaler = None
        self._grads_scaled = False  # True between a scaled backward and the step that consumes (or the unscale_ that rescales) it
        from alpro_amd import amp
        if amp.needs_loss_scaling():
            amp.scaler_for(self)

    # ---- flat buffers ---------------------------------------------------------------------------------
    def _build(self):
        seen, live = set(), []
        for p in self.params:
            if real_grad(p) is not None and id(p) not in seen:
                seen.add(id(p))
                live.append(p)
        if not live:
            return False
        # matrices first (original order), vectors after: BERT's query/key/value weights (and their biases) then sit back
        # to back in the flat gradient buffer, so the fused-QKV weight gradient is ONE N=2304 GEMM into a (2304, 768) view
        live.sort(key=lambda p: p.dim() < 2)
        dev = live[0].device
        offs, n = [], 0
        for p in live:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        fp = torch.zeros(n, dtype=torch.float32, device=dev)
        fg = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(live, offs):
                k = p.numel()
                fp[o:o + k].copy_(p.data.reshape(-1))
                fg[o:o + k].copy_(p.grad.reshape(-1))
                p.data = fp[o:o + k].view(p.shape)
                p.grad = fg[o:o + k].view(p.shape)
        self.flat = dict(p=fp, g=fg, m=torch.zeros_like(fp), v=torch.zeros_like(fp), n=n, live=live, offs=offs,
                         norm=torch.zeros(1, dtype=torch.float32, device=dev))
        bump_param_epoch()
        register_flat_lp(fp, None, live)  # announces the flat range (weights.param_version); the 16-bit mirror follows in step()
        self._span = {id(p): (o, o + (p.numel() + 3) // 4 * 4) for p, o in zip(live, offs)}
        # "step() left the gradient buffer zeroed" (fused_zero_grad) holds until something writes a gradient: the hand-written backward passes
        # announce themselves (train.BACKWARD_EPOCH), torch's own accumulation (heads, temperature: a handful of tensors) through this hook --
        # a driver that calls scaled_loss.backward() itself and then discards the step with zero_grad() must get a real clear;
        # the hooks are registered by the constructor
        if self.allreduce and self.overlap_backward and dist.collectives_active():
            dist.register_grads_final_hook(self._on_grads_final)
        if self._pending_state is not None:
            pend, self._pending_state = self._pending_state, None
            self._restore_moments(pend)
        return True

    def _mark_dirty(self, _p=None):
        self._g_clean = False
        self._writes += 1

    def _stamp(self):
        return (self._writes, _backward_epoch())

    def _note_synced(self, kind):
        """kind: "sum" / "avg" (the gradients as they stand are exchanged) or None."""
        self._pre_synced = kind
        self._sync_stamp = self._stamp() if kind is not None else None

    def _sync_is_current(self):
        return self._pre_synced is not None and self._sync_stamp == self._stamp()

    def _layout(self):
        """[(index of the parameter in the constructor's list, flat offset, numel)]: what the m / v buffers mean."""
        pos = {id(p): i for i, p in enumerate(self.params)}
        return [(pos[id(p)], o, p.numel()) for p, o in zip(self.flat["live"], self.flat["offs"])]

    def _restore_moments(self, sd):
        lay = [tuple(int(x) for x in e) for e in sd["layout"]]
        if lay != self._layout():
            raise RuntimeError("FlatAdamW.load_state_dict: the checkpoint's flat layout (%d tensors) does not match this model's (%d): "
                               "different set of trained parameters" % (len(lay), len(self.flat["live"])))
        for k in ("m", "v"):
            self.flat[k].copy_(sd[k].to(self.flat[k].device, torch.float32).reshape(-1))

    def master_params(self):
        """Objects carrying the gradients this optimizer will apply (for gradient clipping by the caller): the flat view once it exists."""
        if self.flat is not None:
            return [_FlatView(self.flat["g"])]
        return [p for p in self.params if real_grad(p) is not None]

    @property
    def n_params(self):
        return 0 if self.flat is None else sum(p.numel() for p in self.flat["live"])

    def zero_grad(self, set_to_none=False):
        for h, _, _ in self._inflight:  # a driver that skips step() (e.g. on a NaN loss) must not zero under a running all-reduce
            h.wait()
        self._inflight, self._reduced = [], []
        self._reserve_cus(False)
        self._note_synced(None)
        if self.flat is None:
            for p in self.params:
                p.grad = None
        elif self._g_clean and self._clean_epoch == _backward_epoch():
            self._g_clean = False      # step() already cleared the buffer (fused_zero_grad) and no backward has written into it since
        else:
            self._g_clean = False
            self.flat["g"].zero_()

    # ---- overlapped exchange -----------------------------------------------------------------------------
    @staticmethod
    def _merge(ranges):
        out = []
        for s, e in sorted(ranges):
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return [(s, e) for s, e in out]

    def _ranges_of(self, params=None, all_but=None):
        if params is not None:
            spans = [self._span[id(p)] for p in params if id(p) in self._span]
        else:
            skip = {id(p) for p in all_but}
            spans = [v for k, v in self._span.items() if k not in skip]
        return self._merge(spans)

    def _reserve_cus(self, on):
        """While all-reduces launched from inside backward are in flight, the launches of THIS stream (the one backward runs on) plan for
        dist.rccl_cu_reserve() fewer CUs: the weight-gradient GEMM's token-range plan is one unit per CU and cannot give a displaced unit to
        anybody else, and the persistent NT grid leaves RCCL's channels room to start (round 5: the NT GEMM itself no longer needs this -- its
        dynamic tile scheduler survives a taken CU --, and the knob is a per-stream override, not process-wide state)."""
        r = dist.rccl_cu_reserve()
        if r > 0 and on != self._cus_reserved:
            if on:
                ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count if torch.cuda.is_available() else 256
                self._budget_seen = max(64, ncu - r)
                self._cu_stream = hip.set_stream_option("cu_budget", max(64, ncu - r))
                # ... and the side stream this launch stream sends its weight-gradient GEMMs to (alpro_amd.config, ALPRO_WGRAD_STREAM): the kernel that
                # needs the budget most -- one workgroup per CU, a displaced one waits a whole round -- runs there
                import ctypes
                from alpro_amd import config as rt
                on_gpu = torch.cuda.is_available() and self.flat is not None and self.flat["g"].is_cuda
                self._cu_side = [ctypes.c_void_p(h) for h in rt.side_streams_of_current(self.flat["g"].device)] if on_gpu else []
                for h in self._cu_side:
                    hip.set_stream_option("cu_budget", max(64, ncu - r), stream=h)
            else:
                hip.set_stream_option("cu_budget", -1, stream=getattr(self, "_cu_stream", None))
                for h in getattr(self, "_cu_side", []):
                    hip.set_stream_option("cu_budget", -1, stream=h)
                self._cu_side = []
            self._cus_reserved = on

    def _launch(self, s, e):
        if self.overlap_backward:
            self._reserve_cus(True)
        g = self.flat["g"][s:e]
        if self.wire_dtype is not None and self.wire_dtype != torch.float32:
            w = g.to(self.wire_dtype)
            h = torch.distributed.all_reduce(w, async_op=True)
            self._inflight.append((h, w, g))
        else:
            h = torch.distributed.all_reduce(g, async_op=True)
            self._inflight.append((h, None, g))
        self._reduced.append((s, e))

    def _on_grads_final(self, params=None, all_but=None):
        self._writes += 1
        self._g_clean = False   # gradients are being written (a driver's own scaled_loss.backward() does not pass through backward(): ADVICE r4)
        if self.flat is None or not self.allreduce or not dist.collectives_active():
            return
        if all_but is not None and not any(id(p) in self._span for p in all_but):
            return  # "everything but <module>" from a module this optimizer does not train (another model in the same process)
        done = self._merge(self._reduced)
        for s, e in self._ranges_of(params, all_but):
            if any(s < de and ds < e for ds, de in done):
                raise RuntimeError("FlatAdamW: a gradient range was reported final twice in one step (several backward passes per step?); "
                                   "construct the optimizer with overlap_backward=False for gradient accumulation")
            for c in range(s, e, self.bucket_elems):
                self._launch(c, min(e, c + self.bucket_elems))
                self._early[0] += 1
                self._early[1] += min(e, c + self.bucket_elems) - c

    def _finish_exchange(self):
        """All-reduce whatever is not on the wire yet, then make the compute stream wait for every handle."""
        n = self.flat["n"]
        rec = self.record_exchange
        on_gpu = rec and self.flat["g"].is_cuda
        if rec:
            import time
            t0 = time.perf_counter()
            if on_gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
        late = [0, 0]
        pos = 0
        for s, e in self._merge(self._reduced) + [(n, n)]:
            for c in range(pos, s, self.bucket_elems):
                self._launch(c, min(s, c + self.bucket_elems))
                late[0] += 1
                late[1] += min(s, c + self.bucket_elems) - c
            pos = max(pos, e)
        for h, w, g in self._inflight:
            h.wait()
            if w is not None:
                g.copy_(w)
        self._inflight, self._reduced = [], []
        if rec:
            if on_gpu:
                e1.record()
            wire = 4 if self.wire_dtype in (None, torch.float32) else torch.empty(0, dtype=self.wire_dtype).element_size()
            self.exchange_log.append(dict(ranges_on_wire_early=self._early[0], bytes_on_wire_early=self._early[1] * wire, ranges_at_synchronize=late[0],
                                          bytes_at_synchronize=late[1] * wire, cu_budget=self._budget_seen if self._cus_reserved else None,
                                          events=(e0, e1) if on_gpu else None, host_ms=(time.perf_counter() - t0) * 1e3))
        self._early = [0, 0]
        self._reserve_cus(False)

    def exchange_stats(self, last=None):
        """Mean over the logged exchanges (the last `last` of them): what the N > 1 bench line prints.  comm_exposed_ms = device time between the
        moment backward had returned and synchronize() / step() asked for the sums, and the moment the compute stream got them (late launches +
        waits for every handle): exchange time NOT hidden under backward.  With CPU tensors (gloo tests) the host time of the same span."""
        log = self.exchange_log[-last:] if last else self.exchange_log
        if not log:
            return None
        ms = []
        for r in log:
            if r["events"] is not None:
                r["events"][1].synchronize()
                ms.append(r["events"][0].elapsed_time(r["events"][1]))
            else:
                ms.append(r["host_ms"])
        k = float(len(log))
        budgets = sorted({r["cu_budget"] for r in log if r["cu_budget"] is not None})
        return dict(exchanges=len(log), comm_exposed_ms=round(sum(ms) / k, 3), comm_exposed_ms_max=round(max(ms), 3),
                    ranges_on_wire_early=round(sum(r["ranges_on_wire_early"] for r in log) / k, 2), bytes_on_wire_early=int(sum(r["bytes_on_wire_early"] for r in log) / k),
                    ranges_at_synchronize=round(sum(r["ranges_at_synchronize"] for r in log) / k, 2), bytes_at_synchronize=int(sum(r["bytes_at_synchronize"] for r in log) / k),
                    cu_budget_while_in_flight=budgets[0] if len(budgets) == 1 else (budgets or None), overlap_backward=bool(self.overlap_backward),
                    timer="device events on the compute stream" if log[0]["events"] is not None else "host clock (CPU tensors)")

    def synchronize(self, average=False):
        """Sum gradients across ranks; step() folds the 1/world averaging into the AdamW kernel's grad_scale.
        average=True (the hvd.DistributedOptimizer.synchronize() contract: callers clip the AVERAGED gradients before
        step()) divides in place instead and makes the next step() skip both its own exchange and the scaling."""
        if not self.allreduce or not dist.collectives_active():
            self._note_synced("avg" if average else None)
            return 0
        if self._sync_is_current() and not self._inflight:
            # the exchange of the gradients AS THEY STAND already happened -- amp.unscale_ finished an overlapped exchange that was still in
            # flight ("sum"), or the caller synchronised twice without a backward in between (a facade that lost track: ADVICE r4).  A second
            # all-reduce would multiply the gradients by world once more; only the averaging may still be owed.
            if average and self._pre_synced == "sum":
                if self.flat is not None:
                    self.flat["g"].div_(dist.size())
                else:
                    for p in self.params:
                        if real_grad(p) is not None:
                            p.grad.div_(dist.size())
                self._note_synced("avg")
            return 0
        if self._pre_synced == "sum":
            # a backward has accumulated LOCAL gradients on top of an exchanged SUM: sum1 + g2_local cannot be turned into sum1 + sum2 by
            # any collective (an all-reduce gives world * sum1 + sum2).  On top of an exchanged AVERAGE it can (below) -- that is the
            # reference's gradient-accumulation pattern, every micro-step followed by the facade's synchronize(average=True)
            raise RuntimeError("FlatAdamW.synchronize: gradients were accumulated on top of an already exchanged SUM; with several backward "
                               "passes per step call synchronize(average=True) after each of them (hvd.DistributedOptimizer.synchronize does) "
                               "or only once before step()")
        # _pre_synced == "avg" with newer gradients on top (micro-step k of an accumulation window: the buffer holds avg_1 + .. + avg_(k-1)
        # + g_k local): every rank holds the same averaged part, so the all-reduce returns world * (avg_1 + ..) + sum_k = sum_1 + .. + sum_k,
        # the plain SUM again -- and the average after the division, exactly what Horovod's repeated averaging all-reduce leaves behind
        if self.flat is None:  # first step: gradients are still separate tensors
            dist.allreduce_grads_(self.params, average=average)
            self._note_synced("avg" if average else "sum")
            return 0
        g, n = self.flat["g"], self.flat["n"]
        self._finish_exchange()
        if average:
            g.div_(dist.size())
        self._note_synced("avg" if average else "sum")
        return n * 4

    def backward(self, loss):
        self._g_clean = False
        self._writes += 1
        return self._backward(loss)

    def _backward(self, loss):
        """loss.backward() for this optimizer's parameters; with fp16 operands the loss is first multiplied by the (device-resident) loss
        scale and step() divides it out again inside the AdamW kernel -- the fused form of apex's
        `with amp.scale_loss(loss, optimizer, delay_unscale=True) as s: s.backward()` (run_pretrain_sparse.py:596-599)."""
        from alpro_amd import amp
        if not amp.needs_loss_scaling():
            return loss.backward()
        sc = amp.scaler_for(self).to(loss.device)
        self._grads_scaled = True
        with rt.loss_scaling(sc):
            (loss * sc.scale.reshape(())).backward()

    def step(self, closure=None):
        if self._pre_synced is not None and not (self._sync_is_current() and not self._inflight) and self.allreduce and dist.collectives_active():
            # gradients were written after the last synchronize() (an accumulation micro-step that was not followed by one): exchange them now
            self.synchronize(average=self._pre_synced == "avg")
        pre = self._pre_synced
        self._note_synced(None)
        if self.flat is None and not self._build():
            # no parameter has a gradient yet: a no-op like torch.optim (the reference calls optimizer.step() once before the
            # first backward, run_pretrain_sparse.py:508-511)
            return None
        f, grp = self.flat, self.param_groups[0]
        late = [p for p in self.params if real_grad(p) is not None and (p.grad.data_ptr() < f["g"].data_ptr() or p.grad.data_ptr() >= f["g"].data_ptr() + f["n"] * 4)]
        if late:
            raise RuntimeError("%d parameters started receiving gradients after the flat buffers were built" % len(late))
        if pre is None:
            self.synchronize()
            self._note_synced(None)
        world = dist.size() if (self.allreduce and pre != "avg") else 1
        if self.scaler is not None and getattr(self, "_scaler_steps_synced", None) is not self.scaler:
            # a loss scaler attached after some plain steps, or after load_state_dict(): the DEVICE step counter that drives Adam's bias
            # correction under loss scaling continues from the host counter instead of restarting at zero
            self.scaler.to(f["g"].device).state[2] = float(self.step_count)
            self._scaler_steps_synced = self.scaler
        self.step_count += 1
        lr, (b1, b2) = grp["lr"], grp["betas"]
        step_size = lr
        if grp["correct_bias"]:
            step_size = lr * math.sqrt(1.0 - b2 ** self.step_count) / (1.0 - b1 ** self.step_count)
        norm = None
        sc = self.scaler
        if sc is not None:
            sc.to(f["g"].device)
        if (self.max_grad_norm is not None and self.max_grad_norm > 0) or sc is not None:
            norm = f["norm"]
            norm.zero_()
            hip.sumsq(f["g"], norm)
            self.last_grad_norm = norm  # squared norm of the SUMMED (and, after a scaled backward, loss-scaled) gradient, device tensor (no host sync)
        # the 16-bit mirror of the parameters (the GEMM operands are views of it, weights.py) is refreshed BY the optimizer pass (round 6: it was a
        # separate cast launch over the same 0.94 GB); allocated -- and filled once, so that an overflow-skipped first step leaves a valid mirror -- here
        dt = rt.compute_dtype()
        lp = None
        if dt != torch.float32:
            if f.get("lp") is None or f["lp"].dtype != dt:
                f["lp"] = torch.empty(f["n"], dtype=dt, device=f["p"].device)
                hip.cast(f["p"], dt, out=f["lp"])
            lp = f["lp"]
        hip.adamw_step(f["p"], f["g"], f["m"], f["v"], lr, b1, b2, grp["eps"], grp["weight_decay"], step_size, norm,
                       float(self.max_grad_norm or 0.0), 1.0 / world, dyn_state=sc.state if sc is not None else None,
                       grads_scaled=self._grads_scaled, correct_bias=grp["correct_bias"], zero_grad=self.fused_zero_grad, lp=lp)
        self._g_clean = self.fused_zero_grad
        self._clean_epoch = _backward_epoch()
        if sc is not None:  # overflow -> the kernel skipped the update; the schedule halves / grows the scale on the device
            dyn = sc.dynamic
            hip.loss_scale_update(sc.state, norm, sc.growth if dyn else 1.0, sc.backoff if dyn else 1.0, sc.window, sc.min_scale, sc.max_scale)
        self._grads_scaled = False
        bump_param_epoch()
        if lp is not None:  # the 16-bit GEMM operands of every parameter are current again (weights.py)
            register_flat_lp(f["p"], lp, f["live"])
        from alpro_amd.modeling.train import refresh_transposed_operands
        refresh_transposed_operands()  # the dgrad operands W^T of every Linear, one launch (modeling/train.py)

    def state_dict(self):
        """{'step', 'param_groups', 'layout', 'm', 'v'}: the Adam moments as the two flat buffers plus the layout that gives them
        meaning (parameter index in the constructor's list, offset, numel).  Savers may move / narrow the tensors
        (E2E_TrainingRestorer stores fp16 on the CPU, load_save.py:181-196); load_state_dict widens them again."""
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        # under loss scaling the bias correction runs on the scaler's DEVICE counter of APPLIED steps (overflow-skipped steps do not count,
        # the host `step` does): the scaler state travels with the optimizer so that a resumed run continues bit for bit
        extra = dict(loss_scaler=self.scaler.state_dict()) if self.scaler is not None else {}
        if self.flat is None:
            if self._pending_state is not None:
                return dict(self._pending_state, step=self.step_count, param_groups=groups, **extra)
            return dict(step=self.step_count, param_groups=groups, layout=[], m=None, v=None, **extra)
        return dict(step=self.step_count, param_groups=groups, layout=self._layout(), m=self.flat["m"], v=self.flat["v"], **extra)

    def load_state_dict(self, sd):
        """Restore the step counter, hyper-parameters and moments written by state_dict().  The flat buffers only exist after the
        first backward + step(): before that the moments are parked and applied by the first step(), which checks that the same set
        of parameters is being trained (the layout is a pure function of which parameters receive gradients)."""
        self.step_count = int(sd["step"])
        self._scaler_steps_synced = None   # re-seed the device step counter from the restored host counter at the next step() ...
        if sd.get("loss_scaler") is not None:   # ... unless the checkpoint carries the scaler itself: then its applied-step count is authoritative
            from alpro_amd import amp
            sc = amp.scaler_for(self, create=amp.needs_loss_scaling())
            if sc is not None:
                sc.load_state_dict(sd["loss_scaler"])
                self._scaler_steps_synced = sc
        for g, saved in zip(self.param_groups, sd.get("param_groups", [])):
            for k, v in saved.items():
                g[k] = tuple(v) if k == "betas" else v
        if sd.get("m") is None:
            self._pending_state = None
            if self.flat is not None:
                self.flat["m"].zero_()
                self.flat["v"].zero_()
            return
        if self.flat is None:
            self._pending_state = dict(layout=[tuple(e) for e in sd["layout"]], m=sd["m"], v=sd["v"])
        else:
            self._restore_moments(sd)
