"""GPU: device-side batch preparation -- alpro_prepare_clips (ImageNorm + MPM random erase in one pass) and the
batched torch ops of alpro_amd/input_gpu.py.  Pinned to the reference since round 3: tests/golden/input_pipeline_B4_T2.npz holds the
outputs of the reference's OWN random_erase / ImageNorm / mask_batch_text_tokens (ast-extracted from dataset_pretrain_sparse.py:277-311
and data_utils.py:23-70,437-457 and executed under recorded seeds by tests/golden/make_golden.py); the per-sample restatement below
stays as a second, size-independent checker."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _reference_batch(raw_f32, boxes, patch):
    """Per-sample random_erase with a fixed rectangle + ImageNorm of visual / crop / context, all on the device in fp32."""
    crops, ctxs, masks = [], [], []
    H, W = raw_f32.shape[-2:]
    for b, (top, left, h, w) in enumerate(boxes):
        img = raw_f32[b]
        ctx = img.clone()
        ctx[:, :, top:top + h, left:left + w] = 0
        crop = F.pad(img[:, :, top:top + h, left:left + w], (left, W - left - w, top, H - top - h), mode="constant", value=0.0)
        m = torch.ones_like(crop)
        m[:, :, top:top + h, left:left + w] = 0
        masks.append(F.avg_pool2d(m.float(), kernel_size=(patch, patch), stride=patch).mean((0, 1)))
        crops.append(crop)
        ctxs.append(ctx)
    mean = torch.tensor(MEAN, device=raw_f32.device).view(1, 1, 3, 1, 1)
    std = torch.tensor(STD, device=raw_f32.device).view(1, 1, 3, 1, 1)

    def norm(img):
        img = img.clone()
        if torch.max(img) > 1 and mean.max() <= 1:
            img.div_(255.)
        return img.sub_(mean).div_(std)
    return norm(raw_f32), norm(torch.stack(crops)), norm(torch.stack(ctxs)), torch.stack(masks)


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_prepare_pretrain_clips_matches_per_sample_reference(dtype):
    from alpro_amd.input_gpu import prepare_pretrain_clips, sample_erase_box
    g = torch.Generator().manual_seed(5)
    B, T, H, W = 5, 3, 224, 224
    raw8 = torch.randint(0, 256, (B, T, 3, H, W), generator=g, dtype=torch.uint8).cuda()
    raw = raw8 if dtype == torch.uint8 else raw8.float()
    rng = np.random.RandomState(3)
    boxes = [sample_erase_box(H, W, 16, rng=rng) for _ in range(B)]
    out = prepare_pretrain_clips(raw, MEAN, STD, patch_size=16, boxes=boxes)
    vis, crop, ctx, mask = _reference_batch(raw8.float(), boxes, 16)
    for k, ref in (("visual_inputs", vis), ("crop_visual_inputs", crop), ("context_visual_inputs", ctx)):
        assert out[k].shape == ref.shape and out[k].dtype == torch.float32
        assert float((out[k] - ref).abs().max()) <= 1e-6, k
    assert torch.equal(out["mpm_mask"], mask)
    # erased regions hold the normalised ZERO pixel, not 0 (the erase runs before ImageNorm in the reference)
    top, left, h, w = boxes[0]
    assert abs(float(out["context_visual_inputs"][0, 0, 0, top, left]) - (0 - MEAN[0]) / STD[0]) < 1e-6
    # pixels already in 0..1 are not rescaled (ImageNorm's data-dependent test, data_utils.py:455)
    unit = (raw8.float() / 255.0)
    out1 = prepare_pretrain_clips(unit, MEAN, STD, boxes=boxes)
    assert float((out1["visual_inputs"] - _reference_batch(unit, boxes, 16)[0]).abs().max()) <= 1e-6


def test_device_side_mlm_masking_and_sampled_boxes():
    from alpro_amd.input_gpu import mask_batch_text_tokens, prepare_pretrain_clips
    g = torch.Generator(device="cuda").manual_seed(0)
    ids = torch.randint(1000, 30000, (256, 40), device="cuda", generator=g)
    ids[:, 0], ids[:, 30], ids[:, 31:] = 101, 102, 0
    masked, labels = mask_batch_text_tokens(ids, mask_token_id=103, vocab_size=30522, generator=g)
    sel = labels != -100
    assert masked.is_cuda and not sel[:, 0].any() and not sel[:, 30:].any()
    assert torch.equal(labels[sel], ids[sel]) and torch.equal(masked[~sel], ids[~sel])
    assert 0.12 < float(sel.float().sum() / (256 * 29)) < 0.18 and 0.75 < float((masked[sel] == 103).float().mean()) < 0.85
    raw = torch.randint(0, 256, (8, 2, 3, 224, 224), dtype=torch.uint8, device="cuda")
    out = prepare_pretrain_clips(raw, MEAN, STD, rng=np.random.RandomState(1))
    erased = 1.0 - out["mpm_mask"]
    frac = erased.flatten(1).mean(1)
    assert bool(((frac > 0.2) & (frac < 0.6)).all())           # rectangle area is drawn from [0.3, 0.5] of the image, patch-aligned
    z = torch.tensor([(0 - m) / s for m, s in zip(MEAN, STD)], device="cuda").view(1, 1, 3, 1, 1)
    pix = erased.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, None, None].bool()
    sel = pix.expand_as(out["visual_inputs"])
    assert float((out["context_visual_inputs"][sel] - z.expand_as(out["visual_inputs"])[sel]).abs().max()) < 1e-6
    assert torch.equal(out["crop_visual_inputs"][sel], out["visual_inputs"][sel])


def test_prepare_clips_vs_reference_fixture():
    """alpro_prepare_clips on the fixture's raw uint8 clips, with the rectangles sample_erase_box draws from the recorded numpy seed,
    against what the reference's random_erase + PrefetchLoader ImageNorm produced: strided pixel subsamples of visual / crop / context
    (every 7th row / column), their per-(clip, frame, channel) sums and sums of squares, and mpm_mask."""
    from alpro_amd.input_gpu import prepare_pretrain_clips, sample_erase_box
    from tests.conftest import GOLDEN
    from tests.golden.det_init import det_raw_clips
    g = np.load(os.path.join(GOLDEN, "input_pipeline_B4_T2.npz"))
    B, T = g["visual_inputs_sub"].shape[:2]
    raw = det_raw_clips(B, T).cuda()
    rng = np.random.RandomState(int(g["np_seed"]))
    boxes = [sample_erase_box(224, 224, 16, rng=rng) for _ in range(B)]
    st = int(g["stride"])
    for src in (raw, raw.float()):       # uint8 pixels as the dataloader delivers them, and the .float() the reference makes first
        out = prepare_pretrain_clips(src, MEAN, STD, patch_size=16, boxes=boxes)
        assert np.array_equal(out["mpm_mask"].cpu().numpy(), g["mpm_mask"])
        for k in ("visual_inputs", "crop_visual_inputs", "context_visual_inputs"):
            got = out[k]
            assert float(np.abs(got[..., ::st, ::st].cpu().numpy() - g[k + "_sub"]).max()) <= 2e-6, k
            s1, s2 = got.double().sum((-1, -2)).cpu().numpy(), (got.double() ** 2).sum((-1, -2)).cpu().numpy()
            # (the fixture was produced on the CPU, where img.div_(255.) is a true division; on the device torch -- and this kernel -- multiply
            # by 1/255: 1-ulp differences per pixel, ~1.4e-6 relative on the 50176-pixel sums)
            assert np.allclose(s1, g[k + "_sum"], rtol=1e-5, atol=1e-2) and np.allclose(s2, g[k + "_sqsum"], rtol=1e-5, atol=1e-2), k
    unit = prepare_pretrain_clips(raw.float() / 255.0, MEAN, STD, boxes=boxes)    # 0..1 pixels: no second rescale (data_utils.py:455)
    assert float(np.abs(unit["visual_inputs"][..., ::st, ::st].cpu().numpy() - g["unit_visual_inputs_sub"]).max()) <= 2e-6
