"""`src.optimization.utils.setup_e2e_optimizer` (reference utils.py:5-16), as the unchanged drivers call it (run_pretrain_sparse.py:430,
run_video_retrieval.py): 'adamw' -> the fused flat AdamW of this repo, 'adam' / 'adamax' -> torch's, exactly the reference's choices and
constructor arguments (model.parameters(), lr=opts.learning_rate, betas=opts.betas)."""
from torch.optim import Adam, Adamax

from src.optimization.adamw import AdamW


def setup_e2e_optimizer(model, opts):
    if opts.optim == 'adam':
        OptimCls = Adam
    elif opts.optim == 'adamax':
        OptimCls = Adamax
    elif opts.optim == 'adamw':
        OptimCls = AdamW
    else:
        raise ValueError('invalid optimizer')
    return OptimCls(model.parameters(), lr=opts.learning_rate, betas=opts.betas)
