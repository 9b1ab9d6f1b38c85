"""`horovod.torch` surface used by the reference (run_pretrain_sparse.py:410-439,596-648; run_video_retrieval.py; alpro_models.py:
110-123) on torch.distributed (nccl == RCCL over xGMI on MI355X, gloo on CPU).  Launch one process per GPU with
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ...` instead of `horovodrun -np N`."""
import contextlib

import torch
import torch.distributed as td

from alpro_amd import dist as _d

init = _d.init
rank = _d.rank
size = _d.size
local_rank = _d.local_rank
allgather = _d.allgather
broadcast_parameters = _d.broadcast_parameters


class Compression:
    none = None
    fp16 = None  # accepted and ignored: the flat gradient exchange is fp32


Average, Sum = "average", "sum"


def allreduce_(tensor, average=True, name=None, op=None):
    """In-place all-reduce (sum, or mean when average / op == Average)."""
    if _d.collectives_active():
        td.all_reduce(tensor)
        if (op == Average) or (op is None and average):
            tensor.div_(size())
    return tensor


def allreduce(tensor, average=True, name=None, op=None):
    return allreduce_(tensor.clone(), average=average, name=name, op=op)


def broadcast_(tensor, root_rank, name=None):
    if _d.collectives_active():
        td.broadcast(tensor, src=root_rank)
    return tensor


def broadcast(tensor, root_rank, name=None):
    return broadcast_(tensor.clone(), root_rank, name)


def broadcast_optimizer_state(optimizer, root_rank=0):
    """Every tensor of optimizer.state_dict() (the moments) from root_rank, then the step counter (explicitly, below)."""
    if not _d.collectives_active():
        return
    inner = getattr(optimizer, "_opt", optimizer)

    def walk(o):
        if torch.is_tensor(o):
            td.broadcast(o, src=root_rank)
        elif isinstance(o, dict):
            for k in sorted(o, key=str):
                walk(o[k])
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)

    walk(inner.state_dict())
    if hasattr(inner, "step_count"):  # FlatAdamW keeps its step counter as a Python int: broadcast it explicitly
        dev = next((p.device for g in inner.param_groups for p in g["params"]), torch.device("cpu"))
        t = torch.tensor([int(inner.step_count)], dtype=torch.int64, device=dev)
        td.broadcast(t, src=root_rank)
        inner.step_count = int(t.item())


class _DistributedOptimizer:
    """hvd.DistributedOptimizer: `synchronize()` averages the gradients across ranks, `step()` synchronizes first unless that
    already happened or the call sits inside `skip_synchronize()` (the reference's pattern: backward -> synchronize -> clip
    -> `with skip_synchronize(): step()`).  With alpro_amd.optim.FlatAdamW inside, the exchange is ONE all-reduce of the flat
    gradient buffer; with any other torch optimizer, a few large flat buckets."""

    def __init__(self, optimizer, named_parameters=None):
        self._opt = optimizer
        self._synced = False
        self._skip = False

    @property
    def param_groups(self):
        return self._opt.param_groups

    def _params(self):
        return [p for g in self._opt.param_groups for p in g["params"]]

    def synchronize(self):
        if hasattr(self._opt, "synchronize"):   # FlatAdamW
            self._opt.synchronize(average=True)
        else:
            _d.allreduce_grads_(self._params(), average=True)
        self._synced = True

    @contextlib.contextmanager
    def skip_synchronize(self):
        self._skip = True
        try:
            yield
        finally:
            self._skip = False

    def step(self, closure=None):
        if not self._synced and not self._skip:
            self.synchronize()
        self._synced = False
        out = self._opt.step() if closure is None else self._opt.step(closure)
        if not hasattr(self._opt, "flat"):   # any optimizer but FlatAdamW updated the parameters in place: 16-bit operand copies are stale
            from alpro_amd.modeling.weights import notify_params_updated
            notify_params_updated()
        return out

    def zero_grad(self, *a, **k):
        return self._opt.zero_grad(*a, **k)

    def state_dict(self):
        return self._opt.state_dict()

    def load_state_dict(self, sd):
        return self._opt.load_state_dict(sd)

    def __getattr__(self, name):
        return getattr(self._opt, name)


def DistributedOptimizer(optimizer, named_parameters=None, compression=None, backward_passes_per_step=1, op=Average, **unused):
    return _DistributedOptimizer(optimizer, named_parameters)
