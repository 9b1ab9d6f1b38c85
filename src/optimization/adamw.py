"""`src.optimization.adamw.AdamW` (reference adamw.py:12-103) -> alpro_amd.optim.FlatAdamW: same constructor arguments and defaults
(lr 1e-3, betas (0.9, 0.999), eps 1e-6, weight_decay 0.0, correct_bias True), same update (pinned by tests/golden/optimizer_adamw_3steps.npz),
as two launches over flat buffers instead of ~930 per-tensor Python iterations."""
from alpro_amd.optim import FlatAdamW as AdamW  # noqa: F401
