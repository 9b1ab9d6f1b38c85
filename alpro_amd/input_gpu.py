"""Batch preparation on the device: the three per-batch transforms the reference runs on the host / per
sample -- pixel normalisation, BERT-style MLM masking, and the random-erase crop that produces the MPM inputs -- as batched
tensor ops, so a 300+ pairs/s step is not fed by Python loops over samples and tokenizer calls.

  ImageNorm                      src/datasets/data_utils.py:437-457   (already a device op there; same in-place semantics)
  mask_batch_text_tokens         src/datasets/data_utils.py:23-70     (80 % [MASK] / 10 % random / 10 % kept, specials and padding never masked)
  random_erase_batch             src/datasets/dataset_pretrain_sparse.py:277-311  (rejection-sampled patch-aligned rectangle per sample)

Randomness comes from torch / numpy generators the caller may pass, so runs are reproducible; the sampling DISTRIBUTIONS are
the reference's, the random streams are not (the reference draws per sample on the host).
"""
import numpy as np
import torch


class ImageNorm:
    """(B, N, 3, H, W) float pixels -> (x / 255 if the data is 0..255 and mean <= 1) - mean) / std, in place."""

    def __init__(self, mean, std, device="cuda"):
        self.mean = torch.tensor(mean, dtype=torch.float32, device=device).view(1, 1, 3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32, device=device).view(1, 1, 3, 1, 1)

    def __call__(self, img):
        if torch.max(img) > 1 and self.mean.max() <= 1:
            img.div_(255.)
        return img.sub_(self.mean).div_(self.std)


def mask_batch_text_tokens(inputs, mask_token_id, vocab_size, special_token_ids=(0, 100, 101, 102, 103), pad_token_id=0,
                           mlm_probability=0.15, generator=None):
    """inputs (B, L) int64 on any device, already padded -> (masked inputs, labels) with labels == -100 off the masked positions.
    `special_token_ids` replaces tokenizer.get_special_tokens_mask (bert-base-uncased: [PAD] 0, [UNK] 100, [CLS] 101, [SEP] 102,
    [MASK] 103).  Not in place (the reference overwrites its argument; callers there pass a clone)."""
    inputs = inputs.clone()
    labels = inputs.clone()
    dev = inputs.device
    special = torch.zeros_like(inputs, dtype=torch.bool)
    for t in special_token_ids:
        special |= inputs.eq(t)
    if pad_token_id is not None:
        special |= inputs.eq(pad_token_id)
    prob = torch.full(inputs.shape, mlm_probability, device=dev).masked_fill_(special, 0.0)
    masked = torch.bernoulli(prob, generator=generator).bool()
    labels[~masked] = -100
    replaced = torch.bernoulli(torch.full(inputs.shape, 0.8, device=dev), generator=generator).bool() & masked
    inputs[replaced] = mask_token_id
    rnd = torch.bernoulli(torch.full(inputs.shape, 0.5, device=dev), generator=generator).bool() & masked & ~replaced
    words = torch.randint(vocab_size, inputs.shape, dtype=torch.long, device=dev, generator=generator)
    inputs[rnd] = words[rnd]
    return inputs, labels


def sample_erase_box(img_h, img_w, patch_size, s_l=0.3, s_h=0.5, r_1=0.3, r_2=1 / 0.3, rng=np.random):
    """One patch-aligned rectangle (top, left, h, w) by the reference's rejection sampling."""
    while True:
        s = rng.uniform(s_l, s_h) * img_h * img_w
        r = rng.uniform(r_1, r_2)
        w = int(np.sqrt(s / r))
        h = int(np.sqrt(s * r))
        left = rng.randint(0, img_w)
        top = rng.randint(0, img_h)
        w -= w % patch_size
        h -= h % patch_size
        left -= left % patch_size
        top -= top % patch_size
        if left + w <= img_w and top + h <= img_h:
            return top, left, h, w


def random_erase_batch(visual_inputs, patch_size=16, boxes=None, rng=np.random, **box_kw):
    """visual_inputs (B, T, C, H, W) on the device -> dict(crop_visual_inputs, context_visual_inputs, mpm_mask) as the
    pretraining collator builds them per sample: the crop keeps ONLY the rectangle (zeros elsewhere), the context erases it,
    mpm_mask (B, H/ps, W/ps) is 1 on kept patches and 0 on the rectangle.  One rectangle per sample (host-side sampling of
    4 integers each), applied to the whole batch with two masked selects."""
    B, T, C, H, W = visual_inputs.shape
    if boxes is None:
        boxes = [sample_erase_box(H, W, patch_size, rng=rng, **box_kw) for _ in range(B)]
    dev = visual_inputs.device
    bx = torch.tensor(boxes, dtype=torch.long, device=dev)                      # (B, 4): top, left, h, w
    ys = torch.arange(H, device=dev)[None, :, None]
    xs = torch.arange(W, device=dev)[None, None, :]
    inside = ((ys >= bx[:, 0, None, None]) & (ys < (bx[:, 0] + bx[:, 2])[:, None, None]) &
              (xs >= bx[:, 1, None, None]) & (xs < (bx[:, 1] + bx[:, 3])[:, None, None]))  # (B, H, W)
    m = inside[:, None, None].to(visual_inputs.dtype)
    crop = visual_inputs * m
    context = visual_inputs * (1 - m)
    mpm_mask = 1.0 - torch.nn.functional.avg_pool2d(inside.float()[:, None], kernel_size=patch_size, stride=patch_size)[:, 0]
    return dict(crop_visual_inputs=crop, context_visual_inputs=context, mpm_mask=mpm_mask, boxes=boxes)


def prepare_pretrain_clips(raw, mean, std, patch_size=16, boxes=None, rng=np.random, assume_255=None, **box_kw):
    """The whole visual side of a pretraining batch in ONE kernel (alpro_prepare_clips): raw (B, T, 3, H, W) uint8 / float pixels on
    the device -> dict(visual_inputs, crop_visual_inputs, context_visual_inputs, mpm_mask, boxes), identical to what the reference
    assembles from PretrainCollator's random_erase on raw pixels (dataset_pretrain_sparse.py:277-311) followed by ImageNorm on each of
    the three tensors (dataloader.py:104-115): 1 read + 3 writes instead of ~15 elementwise passes.
    assume_255: True / False fixes ImageNorm's data-dependent `torch.max(img) > 1` test (data_utils.py:455) without a device sync;
    None evaluates it (uint8 input is always 0..255)."""
    from alpro_amd import hip
    B, T, C, H, W = raw.shape
    if boxes is None:
        boxes = [sample_erase_box(H, W, patch_size, rng=rng, **box_kw) for _ in range(B)]
    if assume_255 is None:
        assume_255 = True if raw.dtype == torch.uint8 else bool(torch.max(raw) > 1)
    scale = (1.0 / 255.0) if (assume_255 and max(mean) <= 1) else 1.0
    bx = torch.tensor(boxes, dtype=torch.int32, device=raw.device)
    vis, crop, ctx = hip.prepare_clips(raw.contiguous(), mean, std, scale, boxes=bx)
    gh, gw = H // patch_size, W // patch_size
    ys = torch.arange(gh, device=raw.device)[None, :, None] * patch_size
    xs = torch.arange(gw, device=raw.device)[None, None, :] * patch_size
    b64 = bx.long()
    inside = ((ys >= b64[:, 0, None, None]) & (ys < (b64[:, 0] + b64[:, 2])[:, None, None]) &
              (xs >= b64[:, 1, None, None]) & (xs < (b64[:, 1] + b64[:, 3])[:, None, None]))
    return dict(visual_inputs=vis, crop_visual_inputs=crop, context_visual_inputs=ctx, mpm_mask=1.0 - inside.float(), boxes=boxes)
