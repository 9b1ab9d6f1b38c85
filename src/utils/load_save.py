"""Reference import path `src.utils.load_save` -> the hot-path checkpoint loader (drivers and savers are out of scope)."""
from alpro_amd.utils.load_save import (load_state_dict_with_pos_embed_resizing, resize_spatial_embedding,  # noqa: F401
                                       resize_temporal_embedding)
