"""`src.optimization` under the reference's dotted path: `utils` (setup_e2e_optimizer) and `adamw` (AdamW) resolve to the fused flat optimizer
of this repo, every other module of the package (`sched`: get_lr_sched, ...) to the reference's own file (see src/__init__.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
