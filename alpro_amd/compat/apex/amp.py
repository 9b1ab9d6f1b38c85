"""apex.amp as the reference's drivers call it (run_pretrain_sparse.py:441,596,634; load_save.py:269,336) -> alpro_amd.amp.

The precision policy of the MI355X path is ALPRO_COMPUTE_DTYPE, not apex's opt_level: with bf16 / fp32 operands these are the identity
operations apex itself performs when disabled (`fp16: 0`, every release config); with fp16 operands `scale_loss` scales the loss by a
device-resident dynamic loss scale and the optimizer step unscales, checks for overflow and adapts the scale (alpro_amd/amp.py)."""
from alpro_amd.amp import initialize, load_state_dict, master_params, scale_loss, state_dict  # noqa: F401
