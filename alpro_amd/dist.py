"""Data-parallel communication for the ALPRO path: torch.distributed over RCCL (xGMI) on GPUs,
gloo on CPU for tests.  Exposes the small Horovod surface the reference's model code touches
(`hvd.allgather`, `hvd.rank/size/local_rank`, alpro_models.py:110-123) plus the gradient
all-reduce that replaces hvd.DistributedOptimizer.synchronize (run_pretrain_sparse.py:432,601).

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher).
Single-process use needs no initialisation: size() == 1 and every collective is the identity.
"""
import os
import sys

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


# ALPRO_FORCE_COLLECTIVES=1: a ONE-rank job still creates its process group and sends every exchange through the collective library
# instead of taking the size() == 1 shortcuts.  A 1-GPU box can then execute the RCCL branches (all_gather_into_tensor,
# reduce_scatter_tensor, async all_reduce handles on RCCL's stream, broadcast) that otherwise only run on a multi-GPU node
# (tests/test_dist_gpu.py::test_one_rank_nccl_runs_every_collective).
_FORCE = [os.environ.get("ALPRO_FORCE_COLLECTIVES", "0") == "1"]


def collectives_active():
    """True when the data-path collectives must really be issued: more than one rank, or a forced single-rank group."""
    return is_initialized() and (td.get_world_size() > 1 or _FORCE[0])


def rccl_cu_reserve():
    """Compute units set aside for the collective library's kernels while the gradient exchange overlaps with backward (round 4, measured
    policy: profiles/r4_overlap_cu_contention.txt).  The persistent GEMMs run one workgroup per CU with 128-160 KiB of LDS: a CU that holds an
    RCCL channel's kernel cannot take one, and a 256-workgroup launch then needs a second round for the displaced workgroups -- worse than
    giving the CUs up in the first place.  So (i) RCCL is capped at this many channels (NCCL_MAX_NCHANNELS, unless the user set it) and (ii)
    while async all-reduces are in flight the library sizes its persistent grids for 256 - reserve CUs (option "cu_budget").
    ALPRO_RCCL_CU_RESERVE overrides (0 = no reservation)."""
    return max(0, min(64, int(os.environ.get("ALPRO_RCCL_CU_RESERVE", "16"))))


def _single_node():
    """True only when the launcher SAYS every rank is on this node: torchrun's LOCAL_WORLD_SIZE, Open MPI's OMPI_COMM_WORLD_LOCAL_SIZE or
    Slurm's SLURM_NTASKS_PER_NODE / SLURM_NNODES equal to the world size.  Unknown (a launcher that sets none of them) counts as NOT single
    node: the channel cap below is a single-node xGMI policy and would throttle an inter-node ring."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    for var in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE"):
        if os.environ.get(var, "").isdigit():
            return int(os.environ[var]) == world
    if os.environ.get("SLURM_NNODES", "").isdigit():
        return int(os.environ["SLURM_NNODES"]) == 1
    if os.environ.get("SLURM_NTASKS_PER_NODE", "").isdigit():
        return int(os.environ["SLURM_NTASKS_PER_NODE"]) == world
    return False


def init(backend=None):
    """Initialise from the torchrun-style environment; no-op when WORLD_SIZE is unset or 1 (unless ALPRO_FORCE_COLLECTIVES=1)."""
    if is_initialized() or (int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not _FORCE[0]):
        return
    single_node = _single_node()
    if rccl_cu_reserve() > 0 and os.environ.get("ALPRO_OVERLAP_BACKWARD", "1") != "0" and single_node and "NCCL_MAX_NCHANNELS" not in os.environ:
        # one channel = one resident workgroup = one CU.  Only for the measured configuration (one node, xGMI ring, the overlapped exchange);
        # multi-node jobs and users who set the variable themselves keep RCCL's own choice
        os.environ["NCCL_MAX_NCHANNELS"] = str(rccl_cu_reserve())
        if int(os.environ.get("RANK", "0")) == 0:
            print("alpro_amd.dist: NCCL_MAX_NCHANNELS=%s (ALPRO_RCCL_CU_RESERVE; the overlapped gradient exchange leaves the GEMMs the other CUs)" % os.environ["NCCL_MAX_NCHANNELS"], file=sys.stderr)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend is None:
        # "nccl" IS RCCL on ROCm.  ALPRO_DIST_BACKEND=gloo lets several ranks share one GPU (functional tests of the
        # multi-process path on a single-GPU box; RCCL refuses two ranks on one device).
        backend = os.environ.get("ALPRO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    td.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))


def rank():
    return td.get_rank() if is_initialized() else 0


def size():
    return td.get_world_size() if is_initialized() else 1


def local_rank():
    """Reference quirk kept: VTC target offsets use local_rank(), correct on one node."""
    if is_initialized():
        return int(os.environ.get("LOCAL_RANK", td.get_rank()))
    return 0


# Gradient of the differentiable all-gather.  Horovod 0.19.4 (the reference's pin, env/requirements.txt) implements
# HorovodAllgather.backward as `allreduce(grad_output, average=True)` followed by narrowing to this rank's rows
# (horovod/torch/mpi_ops.py; not vendored under /root/reference -- restated from the published source): the cross-rank VTC key
# gradient is the MEAN over ranks of d loss_r / d feat.  The mathematically exact full-batch gradient is the SUM (what a
# reduce-scatter gives).  Default "average" reproduces the reference's training dynamics; "sum" makes N ranks x B pairs identical to
# one rank x N*B pairs (tests/test_dist_cpu.py checks both).  Select with ALPRO_ALLGATHER_GRAD or set_allgather_grad_mode().
_GATHER_GRAD = [os.environ.get("ALPRO_ALLGATHER_GRAD", "average")]


def set_allgather_grad_mode(mode):
    assert mode in ("average", "sum"), mode
    _GATHER_GRAD[0] = mode


def allgather_grad_mode():
    return _GATHER_GRAD[0]


class _AllGather(torch.autograd.Function):
    """Differentiable all-gather along dim 0: backward reduces the gathered gradient across ranks (mean or sum, see above)
    and keeps this rank's rows -- ONE reduce-scatter on RCCL."""

    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[0]
        out = torch.empty((size() * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        td.all_gather_into_tensor(out, x.contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        if td.get_backend() == "gloo":  # gloo has no reduce_scatter: all-reduce then slice
            g = g.clone()
            td.all_reduce(g)
            out.copy_(g[rank() * ctx.n:(rank() + 1) * ctx.n])
        else:
            td.reduce_scatter_tensor(out, g)
        if _GATHER_GRAD[0] == "average":
            out.div_(size())
        return out


def allgather(x, name=None):
    """Concatenate x from every rank along dim 0, in rank order; gradient flows back (alpro_models.py:110-111)."""
    if not collectives_active():
        return x
    return _AllGather.apply(x)


# ---- "these gradients are final" notifications: the hand-written backward passes tell whoever owns the gradient exchange (FlatAdamW)
# which parameter gradients will not change any more in this step, so their slice of the flat buffer can go onto the wire while the
# rest of backward still runs (Horovod overlaps the same way from its background thread, run_pretrain_sparse.py:432-439).
_GRAD_FINAL_HOOKS = []


def register_grads_final_hook(fn):
    """fn(params=None, all_but=None); kept by weak reference when `fn` is a bound method."""
    import weakref
    ref = weakref.WeakMethod(fn) if hasattr(fn, "__self__") else (lambda f=fn: f)
    _GRAD_FINAL_HOOKS.append(ref)


def grads_final(params=None, all_but=None):
    """Called from backward code: the gradients of `params` (or of every parameter EXCEPT `all_but`) are complete for this step."""
    if not collectives_active() or not _GRAD_FINAL_HOOKS:
        return
    from alpro_amd import config as rt
    rt.join_wgrad()   # "final" includes the weight gradients still queued on the side stream (alpro_amd.config, ALPRO_WGRAD_STREAM)
    rt.join_text_streams()   # ... and the text encoder's, when its backward runs on its own stream (ALPRO_TEXT_STREAM)
    live = []
    for ref in _GRAD_FINAL_HOOKS:
        fn = ref()
        if fn is not None:
            fn(params=params, all_but=all_but)
            live.append(ref)
    _GRAD_FINAL_HOOKS[:] = live


def allreduce_grads_(params, bucket_bytes=64 << 20, average=True):
    """Average gradients across ranks in a few large flat buckets (a ring all-reduce over xGMI is per-link
    bound, so fewer / larger messages win).  Skips parameters without a gradient instead of materialising
    zeros (the reference all-reduces 231 M zero gradients of the frozen prompter).
    Returns the number of bytes reduced."""
    if not collectives_active():
        return 0
    params = list(params)
    real = lambda p: p.grad is not None and (p.grad.dim() == 0 or p.grad.numel() <= 1 or any(p.grad.stride()))   # noqa: E731  (stride-0 placeholders of optim.zero_none_grad: nothing to exchange)
    # Which tensors take part must be the same on every rank (the buckets are positional).  A parameter that received a gradient on some
    # ranks only -- a head or branch that only some batches use -- takes part everywhere: the ranks without one contribute zeros, which is
    # what the reference's zero_none_grad materialises for EVERY such parameter (misc.py:28-31).  One small MAX all-reduce decides.
    if params:
        has = torch.tensor([1 if real(p) else 0 for p in params], dtype=torch.int32, device=params[0].device)
        td.all_reduce(has, op=td.ReduceOp.MAX)
        for p, h in zip(params, has.tolist()):
            if h and not real(p):
                p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params if real(p)]
    sent = 0
    bucket, nbytes = [], 0

    def flush():
        nonlocal bucket, nbytes, sent
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        td.all_reduce(flat)
        if average:
            flat.div_(size())
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        sent += flat.numel() * flat.element_size()
        bucket, nbytes = [], 0

    for g in grads:
        bucket.append(g)
        nbytes += g.numel() * g.element_size()
        if nbytes >= bucket_bytes:
            flush()
    flush()
    return sent


def broadcast_parameters(module_or_state, root_rank=0):
    """hvd.broadcast_parameters (run_pretrain_sparse.py:438): rank 0's parameters/buffers to every rank."""
    if not collectives_active():
        return
    sd = module_or_state.state_dict() if hasattr(module_or_state, "state_dict") else module_or_state
    for _, t in sorted(sd.items()):
        if torch.is_tensor(t):
            td.broadcast(t, src=root_rank)


def barrier():
    if collectives_active():
        td.barrier()
