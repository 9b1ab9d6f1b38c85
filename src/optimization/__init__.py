"""`src.optimization` under the reference's dotted path: `adamw` (AdamW) resolves to the fused flat optimizer of this repo, every other module
of the package to the reference's own file (see src/__init__.py) -- `sched` (get_lr_sched) and `utils`, whose
`from src.optimization.adamw import AdamW` (reference utils.py:3) thereby hands the unchanged drivers' setup_e2e_optimizer the flat optimizer."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
