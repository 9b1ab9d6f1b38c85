"""Checkpoint loading for the hot-path modules: same entry point and semantics as the reference's
`load_state_dict_with_pos_embed_resizing` (src/utils/load_save.py:73-134) and the two resize helpers it calls
(src/modeling/timesformer/helpers.py:355-378).

A released ALPRO / TimeSformer checkpoint carries `pos_embed (1, 1 + P, D)` and `time_embed (1, F, D)` for the grid it was
trained on; loading it into a model with another grid (other resolution / frame count) resamples both tables with
NEAREST-neighbour interpolation along the flattened patch axis / the frame axis, keeps the CLS slot, and then loads every
key whose shape matches (shape-mismatched keys and task heads are skipped, not errors), optionally renaming
`text_encoder.bert.*` -> `text_encoder.*` for the down-stream models.  Everything here is host-side, checkpoint-time work.
"""
import logging

import torch

LOGGER = logging.getLogger(__name__)


def nearest_index(n_in, n_out, device=None):
    """Source index of each of n_out samples under torch's legacy 'nearest' resampling of n_in samples:
    floor(i * n_in / n_out) in fp32, clamped -- the rule F.interpolate(mode='nearest') applies on every axis."""
    scale = torch.tensor(float(n_in) / float(n_out), dtype=torch.float32)
    idx = torch.floor(torch.arange(n_out, dtype=torch.float32) * scale).to(torch.int64)
    return idx.clamp_(max=n_in - 1).to(device) if device is not None else idx.clamp_(max=n_in - 1)


def resize_spatial_embedding(state_dict, key, num_patches):
    """pos_embed (1, 1 + P, D) -> (1, 1 + num_patches, D): CLS slot kept, patch slots resampled along the flattened axis."""
    pos = state_dict[key]
    LOGGER.info("Resizing spatial position embedding from %d to %d", pos.size(1), num_patches + 1)
    idx = nearest_index(pos.size(1) - 1, num_patches, pos.device)
    return torch.cat((pos[:, :1], pos[:, 1:].index_select(1, idx)), 1)


def resize_temporal_embedding(state_dict, key, num_frames):
    """time_embed (1, F, D) -> (1, num_frames, D)."""
    te = state_dict[key]
    LOGGER.info("Resizing temporal position embedding from %d to %d", te.size(1), num_frames)
    return te.index_select(1, nearest_index(te.size(1), num_frames, te.device))


def load_state_dict_with_pos_embed_resizing(model, loaded_state_dict_or_path, num_patches, num_frames,
                                            spatial_embed_key='visual_encoder.model.pos_embed',
                                            temporal_embed_key='visual_encoder.model.time_embed',
                                            strict=False, remove_text_encoder_prefix=False):
    """In place on `model`.  See the module docstring; argument names and defaults follow the reference."""
    if isinstance(loaded_state_dict_or_path, str):
        loaded = torch.load(loaded_state_dict_or_path, map_location="cpu")
    else:
        loaded = loaded_state_dict_or_path
    loaded = dict(loaded)
    if remove_text_encoder_prefix:
        for key in list(loaded):
            if 'text_encoder.bert' in key:
                loaded[key.replace('text_encoder.bert', 'text_encoder')] = loaded.pop(key)
    if num_patches + 1 != loaded[spatial_embed_key].size(1):
        loaded[spatial_embed_key] = resize_spatial_embedding(loaded, spatial_embed_key, num_patches)
    if temporal_embed_key in loaded and num_frames != loaded[temporal_embed_key].size(1):
        loaded[temporal_embed_key] = resize_temporal_embedding(loaded, temporal_embed_key, num_frames)
    own = model.state_dict()
    toload, mismatched = {}, []
    for k, v in own.items():
        if k in loaded:
            if v.shape != loaded[k].shape:
                mismatched.append(k)
            else:
                toload[k] = loaded[k]
    LOGGER.info("Keys in loaded but not in model: %s", sorted(set(loaded) - set(own)))
    LOGGER.info("Keys in model but not in loaded: %s", sorted(set(own) - set(loaded)))
    LOGGER.info("Keys in model and loaded, but shape mismatched: %s", sorted(mismatched))
    model.load_state_dict(toload, strict=strict)
    return mismatched
