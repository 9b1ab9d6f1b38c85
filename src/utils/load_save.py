"""Reference import path `src.utils.load_save` -> the checkpoint loader, savers and restorers of alpro_amd.utils.load_save."""
from alpro_amd.utils.load_save import (E2E_TrainingRestorer, ModelSaver, TrainingRestorer,  # noqa: F401
                                       load_state_dict_with_pos_embed_resizing, resize_spatial_embedding, resize_temporal_embedding)
