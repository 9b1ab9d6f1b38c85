"""Drop-in stand-ins for the two distributed-training packages the reference's drivers import (`horovod.torch`, `apex.amp`),
backed by torch.distributed over RCCL.  Opt in by putting this directory on sys.path BEFORE the driver imports them:

    import alpro_amd.compat, sys; sys.path.insert(0, alpro_amd.compat.PATH)
    from horovod import torch as hvd            # run_pretrain_sparse.py:14 resolves to alpro_amd/compat/horovod/torch
    from apex import amp                         # run_pretrain_sparse.py:12

Only the calls SURVEY.md section 8(b) lists are provided (init/rank/size/local_rank/allgather/allreduce(_)/broadcast(_)/
broadcast_parameters/broadcast_optimizer_state/DistributedOptimizer/Compression.none; amp.initialize/scale_loss/master_params/
state_dict/load_state_dict with fp16 = 0 semantics).
"""
import os

PATH = os.path.dirname(os.path.abspath(__file__))
