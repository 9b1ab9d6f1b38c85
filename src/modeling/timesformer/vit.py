"""`src.modeling.timesformer.vit` -> alpro_amd.modeling.timesformer.vit."""
from alpro_amd.modeling.timesformer.vit import (Attention, Block, DropPath, Mlp, PatchEmbed, TimeSformer,  # noqa: F401
                                                VisionTransformer, trunc_normal_)
