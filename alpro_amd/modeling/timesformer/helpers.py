"""Checkpoint-time key remapping for the TimeSformer visual encoder.

`TimeSformer.load_state_dict(path)` is how the reference's `load_separate_ckpt` initialises the visual encoder
(alpro_models.py:45-51,375-387 -> vit.py:515-533): the string selects one of three sources
(src/modeling/timesformer/helpers.py:207-375)

  "vit_base_patch16_224"   ImageNet ViT-B/16 (timm download in the reference)
  "...CLIP_ViT..."         a CLIP ViT-B/16 state_dict file
  anything else            a Kinetics-pretrained TimeSformer checkpoint file ('model_state' / 'state_dict' / raw dict,
                           'model.' / 'module.' key prefixes stripped)

and all three are remapped onto `VisionTransformer`'s keys: image checkpoints have no temporal branch, so
`blocks.i.temporal_attn.*` / `blocks.i.temporal_norm1.*` start as copies of `blocks.i.attn.*` / `blocks.i.norm1.*`; pos / time
tables are resampled (nearest) when the grid or the frame count differs; shape-mismatched keys (the classifier) are
skipped.  Host-side, runs once per job; nothing here touches the GPU path.

There is no network on the GPU boxes and `timm` is not installed: the ImageNet source is read from a local file named by
`ALPRO_VIT_IMAGENET_CKPT` (a timm `vit_base_patch16_224` state_dict saved with torch.save) and fails loudly otherwise.
"""
import logging
import os
from collections import OrderedDict

import torch

from alpro_amd.utils.load_save import resize_spatial_embedding, resize_temporal_embedding

LOGGER = logging.getLogger(__name__)


def read_checkpoint(path, use_ema=False):
    """File -> flat state_dict with the wrapper levels removed (helpers.py:26-51): `state_dict[_ema]` (strip 'module.'),
    `model_state` (strip 'model.'), a top-level 'model' entry, or the dict itself."""
    if not (path and os.path.isfile(path)):
        raise FileNotFoundError("No checkpoint found at %r" % (path,))
    try:
        ckpt = torch.load(path, map_location="cpu")                       # tensors-only unpickling (torch >= 2.6 default)
    except Exception as e:  # noqa: BLE001 -- released TimeSformer .pyth files also pickle their config objects
        LOGGER.warning("%s is not a tensors-only checkpoint (%s); loading it with full unpickling -- only do this for files you trust", path, type(e).__name__)
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(ckpt, dict):
        key = "state_dict_ema" if (use_ema and "state_dict_ema" in ckpt) else "state_dict"
        if key in ckpt:
            return OrderedDict((k[7:] if k.startswith("module") else k, v) for k, v in ckpt[key].items())
        if "model_state" in ckpt:
            return OrderedDict((k[6:] if k.startswith("model") else k, v) for k, v in ckpt["model_state"].items())
        if "model" in ckpt and isinstance(ckpt["model"], dict):
            return OrderedDict(ckpt["model"])
    return ckpt


def seed_temporal_branch(sd):
    """Image checkpoints carry only the spatial branch: every `blocks.*.attn.*` / `blocks.*.norm1.*` entry is duplicated under
    the temporal name unless the checkpoint already has it (helpers.py:187-203,223-238,311-326)."""
    out = OrderedDict(sd)
    for key, val in sd.items():
        if "blocks" not in key:
            continue
        for src, dst in (("attn", "temporal_attn"), ("norm1", "temporal_norm1")):
            if src in key:
                new_key = key.replace(src, dst)
                out[new_key] = sd[new_key] if new_key in sd else val
    return out


def load_matching(model, sd, what):
    """Load every key whose name AND shape match; report the rest (helpers.py:240-262,328-352).  Returns
    (missing_in_checkpoint, unexpected_in_checkpoint, shape_mismatched)."""
    own = model.state_dict()
    take, mismatched = {}, []
    for k, v in own.items():
        if k in sd:
            if tuple(sd[k].shape) == tuple(v.shape):
                take[k] = sd[k]
            else:
                mismatched.append(k)
    missing = sorted(k for k in own if k not in sd)
    unexpected = sorted(k for k in sd if k not in own)
    LOGGER.info("%s: loading %d tensors; keys in checkpoint but not in model: %d %s; in model but not in checkpoint: %d %s; shape "
                "mismatched: %d %s", what, len(take), len(unexpected), unexpected[:8], len(missing), missing[:8], len(mismatched), mismatched[:8])
    torch.nn.Module.load_state_dict(model, take, strict=False)
    return missing, unexpected, mismatched


def _resize_tables(sd, num_patches, num_frames):
    if "pos_embed" in sd and num_patches + 1 != sd["pos_embed"].size(1):
        sd["pos_embed"] = resize_spatial_embedding(sd, "pos_embed", num_patches)
    if "time_embed" in sd and num_frames != sd["time_embed"].size(1):
        sd["time_embed"] = resize_temporal_embedding(sd, "time_embed", num_frames)


def load_pretrained_kinetics(model, pretrained_model, num_frames=8, num_patches=196, **unused):
    """K400/K600 TimeSformer checkpoint -> VisionTransformer (helpers.py:264-301): the classifier of the checkpoint is ignored
    (the model keeps its own head), pos / time tables are resampled to the model's grid, then a STRICT load."""
    assert len(pretrained_model) > 0, "Path to pre-trained Kinetics weights not provided."
    sd = OrderedDict(read_checkpoint(pretrained_model))
    own = model.state_dict()
    for k in ("head.weight", "head.bias"):
        if k in own:
            sd[k] = own[k]
    _resize_tables(sd, num_patches, num_frames)
    torch.nn.Module.load_state_dict(model, sd, strict=True)
    LOGGER.info("Loaded Kinetics pre-trained TimeSformer weights from %s", pretrained_model)


def load_pretrained_imagenet(model, pretrained_model="vit_base_patch16_224", num_frames=8, num_patches=196, **unused):
    """ImageNet ViT-B/16 -> divided space-time blocks (helpers.py:207-262).  The reference downloads the weights through timm;
    here they come from the local file $ALPRO_VIT_IMAGENET_CKPT."""
    path = os.environ.get("ALPRO_VIT_IMAGENET_CKPT", "")
    if not os.path.isfile(path):
        raise FileNotFoundError("load_pretrained_imagenet: set ALPRO_VIT_IMAGENET_CKPT to a local timm vit_base_patch16_224 state_dict "
                                "(no network / timm on this box); got %r" % path)
    sd = OrderedDict(read_checkpoint(path))
    sd.pop("head.weight", None)
    sd.pop("head.bias", None)
    _resize_tables(sd, num_patches, num_frames)
    return load_matching(model, seed_temporal_branch(sd), "ImageNet ViT-B/16")


def load_pretrained_CLIP_ViT(model, pretrained_model, num_frames=8, num_patches=196, **unused):
    """CLIP ViT-B/16 state_dict file (already in timm key layout) -> divided space-time blocks (helpers.py:304-352)."""
    sd = OrderedDict(read_checkpoint(pretrained_model))
    _resize_tables(sd, num_patches, num_frames)
    return load_matching(model, seed_temporal_branch(sd), "CLIP ViT-B/16")


def load_visual_checkpoint(timesformer, pretrained_ckpt_path):
    """Dispatch of TimeSformer.load_state_dict(path) (vit.py:515-533)."""
    if pretrained_ckpt_path == "vit_base_patch16_224":
        fn = load_pretrained_imagenet
    elif "CLIP_ViT" in pretrained_ckpt_path:
        fn = load_pretrained_CLIP_ViT
    else:
        fn = load_pretrained_kinetics
    LOGGER.info("Loading TimeSformer checkpoints from %s", pretrained_ckpt_path)
    return fn(timesformer.model, pretrained_model=pretrained_ckpt_path, num_frames=timesformer.num_frames, num_patches=timesformer.num_patches)
