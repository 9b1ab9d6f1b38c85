"""`horovod.torch.mpi_ops` names the reference imports directly (src/utils/distributed.py:14: `from horovod.torch.mpi_ops import rank,
size`), on the torch.distributed facade of the parent package."""
from alpro_amd.dist import local_rank, rank, size  # noqa: F401
from . import allgather, allreduce, allreduce_, broadcast, broadcast_  # noqa: F401
