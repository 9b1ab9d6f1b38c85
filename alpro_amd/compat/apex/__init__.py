"""`apex` stand-in: only `apex.amp` with the reference's fp16 = 0 configuration (every config_release/*.json)."""
from . import amp  # noqa: F401
