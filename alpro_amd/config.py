"""Runtime knobs of the MI355X path (not part of the reference's JSON surface; all optional)."""
import os

import torch

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16,
           "float16": torch.float16, "f32": torch.float32, "fp32": torch.float32, "float32": torch.float32}

# Storage dtype of GEMM / attention operands.  bf16 is the throughput mode BASELINE.json quotes;
# float32 is the exact mode (fp32 MFMA) that reproduces the reference's fp32 arithmetic.
_compute_dtype = _DTYPES[os.environ.get("ALPRO_COMPUTE_DTYPE", "bf16").lower()]


def compute_dtype():
    return _compute_dtype


def set_compute_dtype(dt):
    global _compute_dtype
    _compute_dtype = _DTYPES[dt.lower()] if isinstance(dt, str) else dt
    return _compute_dtype


class use_compute_dtype:
    """Context manager: `with use_compute_dtype(torch.float32): ...`"""

    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        self.prev = compute_dtype()
        set_compute_dtype(self.dt)

    def __exit__(self, *a):
        set_compute_dtype(self.prev)
