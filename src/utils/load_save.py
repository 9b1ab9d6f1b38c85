"""Reference import path `src.utils.load_save` -> the checkpoint loader, savers and restorers of alpro_amd.utils.load_save.
Every public name the reference's module has is here: the drivers import save_training_meta / ModelSaver / the restorers /
load_state_dict_with_pos_embed_resizing (run_pretrain_sparse.py:20-23, run_video_retrieval.py:27-30), the datasets import LOGGER
from this module (dataset_base.py:13, dataset_video_retrieval.py:8)."""
from alpro_amd.utils.load_save import (LOGGER, E2E_TrainingRestorer, ModelSaver, TrainingRestorer, compare_dict_difference,  # noqa: F401
                                       load_state_dict_with_pos_embed_resizing, resize_spatial_embedding, resize_temporal_embedding,
                                       save_training_meta)
from alpro_amd.utils.load_save import _to_cpu, _to_device as _to_cuda  # noqa: F401  (private helpers of the same names)
