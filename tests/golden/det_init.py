"""Deterministic, RNG-free tensors for parity work (TEST INFRASTRUCTURE ONLY).

Every parameter, buffer and synthetic input used by the parity tests is a closed-form
function of (name, shape) built from exact 64-bit integer hashing, so the reference (run
once in the authoring container by tests/golden/make_golden.py), the CPU oracle and the
HIP path can regenerate bit-identical fp32 values on any box without shipping 465 M
weights.  Nothing under alpro_amd/ imports this file; tests, __graft_entry__.smoke() and bench.py (in-run parity against the committed
reference fixtures) do.  Lives beside the fixtures it regenerates (oracle/det_init.py re-exports it).
"""
import zlib

import numpy as np
import torch

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x):
    x = (x + _GOLD).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * _M1
    x = (x ^ (x >> np.uint64(27))) * _M2
    return x ^ (x >> np.uint64(31))


def unit_uniform(name, numel):
    """float64 array in [-1, 1), exact on every platform (24-bit mantissa payload)."""
    seed = np.uint64((zlib.crc32(name.encode()) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(numel, dtype=np.uint64) + seed)
    return (h >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0


_LN_WEIGHT_SUFFIXES = ("LayerNorm.weight", "norm1.weight", "norm2.weight", "norm.weight")


def det_param(name, shape, dtype=torch.float32):
    """Closed-form value for a state_dict entry `name` of `shape` (see module docstring)."""
    shape = tuple(shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    leaf = name.split(".")[-1]
    if leaf == "position_ids":
        return torch.arange(numel, dtype=torch.long).reshape(shape)
    if leaf == "temp":
        return torch.full(shape, 0.07, dtype=dtype)
    v = unit_uniform(name, numel)
    if name.endswith(_LN_WEIGHT_SUFFIXES):
        v = 1.0 + 0.1 * v
    elif leaf == "bias":
        v = 0.02 * v
    elif leaf in ("video_prompt_feat", "image_prompt_feat"):
        v = v.reshape(shape)
        v = v / np.linalg.norm(v, axis=-1, keepdims=True)
    else:
        v = 0.035 * v
    return torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape)).to(dtype)


# state_dict aliases that share storage in the reference (xbert.py:670-677 ties the MLM
# decoder to the word embeddings and its bias to predictions.bias).
def canonical_name(name):
    name = name.replace("cls.predictions.decoder.weight", "bert.embeddings.word_embeddings.weight")
    name = name.replace("cls.predictions.decoder.bias", "cls.predictions.bias")
    return name


def fill_state_dict_(module):
    """Overwrite every parameter and buffer of `module` in place with det_param values."""
    with torch.no_grad():
        sd = module.state_dict()
        for k, t in sd.items():
            t.copy_(det_param(canonical_name(k), t.shape, t.dtype if t.is_floating_point() else torch.float32).to(t.dtype))
    return module


def det_batch(B, T, Lt=40, img=224, vocab=30522, seed_name="batch", with_mlm=True, with_mpm=True,
              pad_tail=True):
    """Synthetic batch in the reference's collate layout (SURVEY §8b / dataset_pretrain_sparse.py:252-264)."""
    def u(name, *shape):
        return torch.from_numpy(unit_uniform(seed_name + "/" + name, int(np.prod(shape))).astype(np.float32).reshape(shape))

    def ints(name, lo, hi, *shape):
        x = (unit_uniform(seed_name + "/" + name, int(np.prod(shape))) + 1.0) * 0.5
        return torch.from_numpy((lo + np.floor(x * (hi - lo))).astype(np.int64).reshape(shape))

    batch = {}
    batch["visual_inputs"] = 1.7 * u("visual_inputs", B, T, 3, img, img)
    ids = ints("text_input_ids", 1000, 30000, B, Lt)
    ids[:, 0] = 101
    mask = torch.ones(B, Lt, dtype=torch.long)
    if pad_tail:
        for b in range(B):
            n_valid = Lt - (3 * b + 2) % max(Lt - 8, 1)
            mask[b, n_valid:] = 0
            ids[b, n_valid:] = 0
    batch["text_input_ids"] = ids
    batch["text_input_mask"] = mask
    if with_mlm:
        sel = (u("mlm_sel", B, Lt) > 0.7) & (mask > 0)
        sel[:, 0] = False
        sel[:, 1] = True  # at least one masked token per row (quirk 4h: zero masked -> NaN)
        mlm_ids = ids.clone()
        mlm_ids[sel] = 103
        labels = torch.full((B, Lt), -100, dtype=torch.long)
        labels[sel] = ids[sel]
        batch["mlm_text_input_ids"] = mlm_ids
        batch["mlm_labels"] = labels
    if with_mpm:
        g = img // 16
        mpm_mask = torch.ones(B, g, g)
        for b in range(B):
            r0, c0 = (2 + b) % (g - 6), (5 + 3 * b) % (g - 6)
            mpm_mask[b, r0:r0 + 6, c0:c0 + 6] = 0
        batch["mpm_mask"] = mpm_mask
        batch["crop_visual_inputs"] = 1.7 * u("crop_visual_inputs", B, T, 3, img, img)
        pix = mpm_mask.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, None, None]
        batch["context_visual_inputs"] = batch["visual_inputs"] * pix
        batch["type"] = "video"
    return batch


class PromptEncoding:
    """Stand-in for the tokenizer's BatchEncoding that Prompter.build_text_prompts reads (.input_ids / .attention_mask,
    alpro_models.py:450-456)."""

    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask


def det_prompts(E, n_templates, Lp, seed_name):
    """E entities x n_templates prompt sentences of <= Lp tokens, template-major like the reference's prompt list
    (alpro_models.py:470-472 chunks the encoded prompts into n_templates groups of E rows)."""
    n = E * n_templates
    x = (unit_uniform(seed_name + "/ids", n * Lp) + 1.0) * 0.5
    ids = torch.from_numpy((1000 + np.floor(x * 29000)).astype(np.int64).reshape(n, Lp))
    ids[:, 0] = 101
    mask = torch.ones(n, Lp, dtype=torch.long)
    for r in range(n):
        nv = Lp - (5 * r + 1) % (Lp - 4)
        mask[r, nv:] = 0
        ids[r, nv:] = 0
    return PromptEncoding(ids, mask)


def det_raw_clips(B, T, img=224, seed_name="input_raw"):
    """uint8 pixels (B, T, 3, img, img), closed form."""
    x = (unit_uniform(seed_name, B * T * 3 * img * img) + 1.0) * 0.5
    return torch.from_numpy(np.minimum(np.floor(x * 256.0), 255).astype(np.uint8).reshape(B, T, 3, img, img))


def det_caption_ids(B, Lt=40, seed_name="input_ids"):
    x = (unit_uniform(seed_name, B * Lt) + 1.0) * 0.5
    ids = torch.from_numpy((1000 + np.floor(x * 29000)).astype(np.int64).reshape(B, Lt))
    ids[:, 0] = 101
    for b in range(B):
        n_valid = Lt - (5 * b + 3) % (Lt - 10)
        ids[b, n_valid - 1] = 102
        ids[b, n_valid:] = 0
    ids[0, 5] = 100  # an [UNK] inside a caption: special, never masked
    return ids


# ---- optimizer trajectory (tests/golden/optimizer_adamw_3steps.npz: the reference's AdamW + get_lr_sched + clip_grad_norm_, three steps)
OPT_SHAPES = [("enc.layer.0.weight", (37, 64)), ("enc.layer.0.bias", (64,)), ("enc.patch.weight", (8, 3, 4, 4)), ("temp", ()),
              ("enc.layer.1.weight", (64, 37)), ("enc.norm.weight", (37,)), ("head.weight", (5, 63))]
OPT_GRAD_SCALES = (2.4, 0.55, 0.11)   # global gradient norms of about 40, 9 and 1.9: above / between / below the two clip thresholds
OPT_SCENARIOS = {
    # run_pretrain_sparse.py:433 (setup_e2e_optimizer: lr + betas, eps 1e-6, weight_decay 0), :615-634 (lr set, clip at cfg.grad_norm, step)
    "release": dict(lr=1e-4, betas=(0.9, 0.98), weight_decay=0.0, grad_norm=20.0, decay="linear", num_train_steps=10, warmup_ratio=0.1),
    "decay": dict(lr=5e-4, betas=(0.9, 0.999), weight_decay=0.01, grad_norm=5.0, decay="constant", num_train_steps=10, warmup_ratio=0.1),
}


def opt_tensors(kind, step=0):
    """Closed-form parameters (kind 'param') / gradients of `step` (kind 'grad') for the optimizer trajectory; shared with the tests."""
    out = []
    for name, shape in OPT_SHAPES:
        n = int(np.prod(shape)) if len(shape) else 1
        if kind == "param":
            v = 0.05 * unit_uniform("opt/param/" + name, n)
        else:
            v = OPT_GRAD_SCALES[step] * unit_uniform("opt/grad/%d/%s" % (step, name), n)
        out.append(torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape)))
    return out
