"""apex.amp as the reference uses it with `enabled=cfg.fp16` == 0 (run_pretrain_sparse.py:441,596,634; load_save.py:269,336):
the precision policy of the MI355X path is set by ALPRO_COMPUTE_DTYPE (16-bit GEMM/attention operands, fp32 everything else,
no loss scaling needed with bf16), so these are the identity operations apex itself performs when disabled."""
import contextlib


def initialize(models, optimizers=None, enabled=True, opt_level="O1", **unused):
    return models if optimizers is None else (models, optimizers)


@contextlib.contextmanager
def scale_loss(loss, optimizers, delay_unscale=False, **unused):
    yield loss


def master_params(optimizer):
    for group in optimizer.param_groups:
        for p in group["params"]:
            yield p


def state_dict():
    return {}


def load_state_dict(sd):
    return None
