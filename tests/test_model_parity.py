"""GPU: the nn.Module API (alpro_amd.modeling) against golden vectors captured from the reference
(tests/golden/*.npz) and against the CPU oracle, on the same closed-form weights and inputs.

Tolerances (absolute unless noted), each at most ~2x the error measured on MI355X:
  * exact mode (ALPRO_COMPUTE_DTYPE=fp32, fp32 MFMA): VTC logits / ITM scores within 1e-3 of the reference
    -- the bar BASELINE.json's north_star states; observed errors are ~1e-5.
  * fp16 mode (the benchmark dtype since round 3): operands rounded to fp16 (11 bits) at every GEMM / attention input, fp32 everything
    else; VTC logits asserted at the north star's 1e-3 (measured 2.4e-4), gradients through a loss-SCALED backward (alpro_amd.amp).
  * bf16 mode: 8 mantissa bits; measured 8e-3 on VTC logits / 1.7e-2 on gradients, asserted at 1.6e-2 / 4e-2.
"""
import math
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN
from tests.test_host_cpu import VENC, make_cfg

pytestmark = pytest.mark.gpu


def argmax_multinomial(w, n=1, *a, **k):
    return w.argmax(dim=-1, keepdim=True)


def backward(loss, mode):
    """loss.backward() the way each operand dtype needs it; returns the factor the parameter gradients carry.  fp16: a scaled backward
    (alpro_amd.amp: fixed scale 4096 here; training uses the dynamic schedule) -- an unscaled fp16 backward is refused by the path."""
    from alpro_amd import amp, config as rt
    if mode != "fp16":
        loss.backward()
        return 1.0
    sc = amp.LossScaler(init_scale=4096.0, dynamic=False, device=loss.device)
    rt.set_armed_loss_scaler(sc)
    with rt.loss_scaling(sc):
        (loss * sc.scale.reshape(())).backward()
    return 4096.0


def arm_scale(mode, device="cuda"):
    """fp16: attach the fixed test scale BEFORE the forward (the LM head writes its logit gradient at forward time); returns a keep-alive."""
    from alpro_amd import amp, config as rt
    if mode != "fp16":
        rt.set_armed_loss_scaler(None)
        return None
    sc = amp.LossScaler(init_scale=4096.0, dynamic=False, device=device)
    rt.set_armed_loss_scaler(sc)
    return sc


def to_dev(batch):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}


def close(got, ref, atol, rtol=0.0, what=""):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    lim = atol + rtol * np.abs(ref)
    if os.environ.get("ALPRO_PARITY_REPORT"):  # measurement run (tools/parity_report.sh): print every error next to its limit, assert nothing
        print("[parity-report] %s | %s | err %.3e | limit %.1e" % (os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0], what, err.max(), atol))
        return float(err.max())
    assert (err <= lim).all(), "%s: max err %.3e (limit %.1e), ref max %.3e" % (what, err.max(), atol, np.abs(ref).max())
    return float(err.max())


@pytest.fixture(scope="module")
def retrieval(fixture_models):
    m, batch, _ = fixture_models("retrieval_T2")      # AlproForVideoTextRetrieval, 2 frames, 3 pairs, closed-form weights
    return m, batch, np.load(os.path.join(GOLDEN, "retrieval_T2_B3.npz"))


@pytest.fixture(scope="module")
def fixture_models(bert_cfg):
    """name -> (model in eval mode on the device, batch, reference VTC logits) of tests/golden/parity_cases.py, built ONCE per module: every
    test of a fixture shares the model (465 M parameters, closed-form weights: ~12 s to build; round 4 rebuilt it 20 times -- VERDICT r4 item 9).
    All four stay resident (8 GB of 288).  Tests that run a backward reset the gradients first (fresh_grads)."""
    from tests.golden import parity_cases as pc
    built = {}

    def get(name):
        if name not in built:
            built[name] = pc.build_case(name, bert_cfg, VENC, make_cfg, "cuda")
        return built[name]
    return get


def fresh_grads(m):
    for p in m.parameters():
        p.grad = None
    return m


@pytest.mark.parametrize("mode,tol_logit,tol_emb", [("fp32", 1e-3, 1e-3), ("bf16", 1.6e-2, 6e-2), ("fp16", 2e-3, 6e-3)])
def test_retrieval_vs_reference(retrieval, monkeypatch, mode, tol_logit, tol_emb):
    from alpro_amd import config as rt
    m, batch, g = retrieval
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    with rt.use_compute_dtype(mode), torch.no_grad():
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        out = m(batch)
        inf = m.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1], text_input_ids=batch["text_input_ids"],
                                       text_input_mask=batch["text_input_mask"]))
    e = {}
    e["video_embeds"] = close(ve[:, [0, 1, 100, 196]], g["video_embeds_rows"], tol_emb, what="video_embeds rows")
    close(ve.norm(dim=-1), g["video_embeds_rownorm"], tol_emb * 10, what="video_embeds norms")
    e["itc_loss"] = close(out["itc_loss"], g["itc_loss"], tol_logit, what="itc_loss")
    e["itm_loss"] = close(out["itm_loss"], g["itm_loss"], tol_logit, what="itm_loss")
    e["itm_scores"] = close(out["itm_scores"], g["itm_scores"], tol_logit, what="itm_scores")
    e["inf_itc_scores"] = close(inf["itc_scores"], g["inf_itc_scores"], min(tol_logit, 1e-3) if mode != "bf16" else tol_logit,
                                what="VTC logits (1 video x n captions) -- the north star's 1e-3 bar for fp32 and fp16")
    e["inf_logits"] = close(inf["logits"], g["inf_logits"], tol_logit, what="inference ITM logits")
    assert torch.equal(out["itm_labels"].cpu(), torch.from_numpy(g["itm_labels"]).long())
    print("\n[parity %s] max abs errors vs reference:" % mode, {k: "%.2e" % v for k, v in e.items()})


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-3), ("bf16", 3e-2), ("fp16", 4e-3)])
def test_pretrain_forward_vs_reference(fixture_models, monkeypatch, mode, tol):
    """All ten outputs of AlproForPretrain.forward (VTC + VTM + MLM + MPM) at 8 frames."""
    from alpro_amd import config as rt
    g = np.load(os.path.join(GOLDEN, "pretrain_T8_B2.npz"))
    m, batch, _ = fixture_models("pretrain_T8")
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    with rt.use_compute_dtype(mode), torch.no_grad():
        out = m(batch)
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        te, tf = m._forward_text_feats(batch)
        vf = m._video_feat(ve)
    e = {}
    for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "mpm_logits"):
        e[k] = close(out[k], g[k], tol, what=k)
    close(out["mpm_labels"], g["mpm_labels"], tol * 1e-2, what="mpm_labels (soft)")
    e["mlm_scores"] = close(out["mlm_scores"][:, :, ::61], g["mlm_scores_cols"], tol, what="mlm_scores")
    # fp16 (since round 4: 16-bit operands + the CLS rows re-evaluated in fp32, alpro_amd.config.cls_precise): asserted at the north star's
    # 1e-3 on ALL four reference fixtures (test_vtc_logits_meet_the_north_star_bar_on_every_fixture); plain fp16 measured 3.4e-4 .. 1.06e-3
    e["sim_v2t"] = close(vf @ tf.t() / m.temp, g["sim_v2t"], {"fp32": 1e-3, "fp16": 1e-3, "bf16": 1.6e-2}[mode], what="VTC logits")
    e["video_feat"] = close(vf, g["video_feat"], tol, what="video_feat")
    e["text_embeds"] = close(te, g["text_embeds"], tol * (1 if mode == "fp32" else 2), what="text_embeds")
    e["video_embeds"] = close(ve[:, [0, 1, 57, 196]], g["video_embeds_rows"], tol * (1 if mode == "fp32" else 2), what="video_embeds")
    assert torch.equal(out["itm_labels"].cpu(), torch.from_numpy(g["itm_labels"]).long())
    print("\n[pretrain parity %s] max abs errors vs reference:" % mode, {k: "%.2e" % v for k, v in e.items()})


@pytest.mark.parametrize("mode,rtol", [("fp32", 5e-3), ("bf16", 4e-2), ("fp16", 1e-2)])
def test_pretrain_gradients_vs_reference(fixture_models, monkeypatch, mode, rtol):
    """loss = mlm + itm + itc + mpm (run_pretrain_sparse.py:557) backward through the hand-written HIP backward:
    per-parameter gradient norms of all 460 trainable tensors and 15 full gradients vs the reference's autograd."""
    from alpro_amd import config as rt
    g = np.load(os.path.join(GOLDEN, "pretrain_T8_B2.npz"))
    m, batch, _ = fixture_models("pretrain_T8")
    fresh_grads(m)
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    with rt.use_compute_dtype(mode):
        keep = arm_scale(mode)
        out = m(batch)
        loss = out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]
        gs = backward(loss, mode)
        del keep
    close(out["itc_loss"], g["itc_loss"], {"fp32": 1e-3, "fp16": 1e-3, "bf16": 1.6e-2}[mode], what="itc_loss (train graph)")
    pd = dict(m.named_parameters())
    if gs != 1.0:
        for p_ in pd.values():
            if p_.grad is not None:
                p_.grad.div_(gs)
    names = [str(n) for n in g["grad_norm_names"]]
    missing = [n for n in names if pd[n].grad is None]
    assert not missing, "no gradient for %s" % missing[:5]
    extra = [n for n, p in pd.items() if p.grad is not None and n not in names]
    assert not extra, "unexpected gradients (frozen prompter / unused head) %s" % extra[:5]
    got = np.array([float(pd[n].grad.norm()) for n in names])
    ref = g["grad_norms"]
    # `temp` is ONE scalar whose gradient is a cancelling sum over the similarity matrix (0.04 on the released-geometry fixture against ~2
    # on retrieval_T2): its error is measured against the scale of the non-cancelling case, not against its own near-zero value
    rel = np.abs(got - ref) / np.where(np.array([n == "temp" for n in names]), np.maximum(ref, 0.5), np.maximum(ref, 1e-5))
    # key.bias gradients are identically 0 in exact arithmetic (softmax is invariant to a per-query shift): the reference
    # holds ~1e-9 of fp32 noise there, so only an absolute bound is meaningful
    zero_grad = np.array([n.endswith("attention.self.key.bias") for n in names])
    assert got[zero_grad].max() < 1e-3
    rel[zero_grad] = 0.0
    worst = int(rel.argmax())
    print("\n[grad parity %s] worst grad-norm rel err %.2e at %s; median %.2e" % (mode, rel.max(), names[worst], np.median(rel)))
    if os.environ.get("ALPRO_PARITY_REPORT"):
        print("[parity-report] pretrain_gradients[%s] | worst grad-norm rel err | err %.3e | limit %.1e" % (mode, rel.max(), rtol))
        return
    assert rel.max() < rtol, (names[worst], got[worst], ref[worst])
    for k in g.files:
        if k.startswith("grad/"):
            r = g[k].astype(np.float64)
            e = np.abs(pd[k[5:]].grad.float().cpu().numpy().astype(np.float64) - r).max()
            assert e <= rtol * max(np.abs(r).max(), 1e-6) + 1e-7, (k, e, np.abs(r).max())


def test_embed_resamples_pos_and_time_tables_like_the_reference():
    """Input grid != checkpoint grid (vit.py:328-340,350-357): pos_embed is resampled nearest-neighbour in 2-D, time_embed in
    1-D; forward tokens and the gradients scattered back to the tables must equal the torch restatement."""
    import torch.nn.functional as F
    from alpro_amd import config as rt
    from alpro_amd.modeling.timesformer.vit import TimeSformer
    torch.manual_seed(5)
    enc = TimeSformer(dict(VENC, num_frm=8), input_format="RGB").cuda()
    m = enc.model
    with torch.no_grad():
        m.pos_embed.normal_(0, 0.5)
        m.time_embed.normal_(0, 0.5)
    B, T, Hh, Ww = 2, 4, 160, 192          # 10 x 12 patches, 4 frames against a 14 x 14 / 8-frame table
    x = torch.randn(B, 3, T, Hh, Ww, device="cuda")
    with rt.use_compute_dtype(torch.float32), torch.no_grad():
        tok, T_, Wg, N = m._embed(x)
    assert (T_, Wg, N) == (T, 12, 120)
    D = m.embed_dim
    pos, te = m.pos_embed.detach().clone().requires_grad_(True), m.time_embed.detach().clone().requires_grad_(True)
    conv = F.conv2d(x.transpose(1, 2).reshape(B * T, 3, Hh, Ww), m.patch_embed.proj.weight.detach(), m.patch_embed.proj.bias.detach(), stride=16)
    pt = conv.flatten(2).transpose(1, 2)                                                  # (B*T, N, D)
    new_pos = F.interpolate(pos[0, 1:].t().reshape(1, D, 14, 14), size=(10, 12), mode="nearest").flatten(2).transpose(1, 2)
    pt = pt + new_pos
    pt = pt.view(B, T, N, D).permute(0, 2, 1, 3)                                           # (B, N, T, D)
    pt = pt + F.interpolate(te.transpose(1, 2), size=T, mode="nearest").transpose(1, 2)[:, None]
    ref = torch.cat([(m.cls_token.detach() + pos[:, :1]).expand(B, 1, D), pt.reshape(B, N * T, D)], 1)
    assert (tok - ref.detach()).abs().max().item() < 2e-4
    dtok = torch.randn_like(tok)
    ref.backward(dtok)
    for p in (m.pos_embed, m.time_embed, m.patch_embed.proj.weight, m.patch_embed.proj.bias, m.cls_token):
        p.grad = None
    with rt.use_compute_dtype(torch.float32), torch.no_grad():
        m._embed_backward(m._last_rows, dtok, B, T, N, Wg)
    assert (m.pos_embed.grad - pos.grad).abs().max().item() < 2e-3
    assert (m.time_embed.grad - te.grad).abs().max().item() < 2e-3


def test_cached_retrieval_eval_equals_forward_inference(retrieval):
    """alpro_amd.retrieval_eval (every video / caption encoded once) reproduces the records the reference's loop builds from
    forward_inference per (video, caption mini-batch) (run_video_retrieval.py:642-690)."""
    from alpro_amd import config as rt
    from alpro_amd.retrieval_eval import eval_retrieval, inference_retrieval_cached
    m, batch, _ = retrieval
    vids = [("v%d" % i, batch["visual_inputs"][i:i + 1]) for i in range(3)]
    ids, mask = batch["text_input_ids"], batch["text_input_mask"]
    cap_ids = ["t%d" % i for i in range(ids.shape[0])]
    with rt.use_compute_dtype("fp32"), torch.no_grad():
        got = inference_retrieval_cached(m, vids, ids, mask, cap_ids, eval_bsz=2)
        ref = []
        for vid_id, v in vids:
            for i in range(0, ids.shape[0], 2):
                out = m.forward_inference(dict(visual_inputs=v, text_input_ids=ids[i:i + 2], text_input_mask=mask[i:i + 2]))
                probs = torch.softmax(out["logits"].float(), 1)[:, 1].tolist()
                sims = out["itc_scores"].float().reshape(-1).tolist()
                ref += [dict(vid_id=vid_id, txt_id=c, score=round(p, 4), sim=round(s, 4)) for c, p, s in zip(cap_ids[i:i + 2], probs, sims)]
    assert [(d["vid_id"], d["txt_id"]) for d in got] == [(d["vid_id"], d["txt_id"]) for d in ref]
    assert max(abs(a["score"] - b["score"]) for a, b in zip(got, ref)) <= 2e-4
    assert max(abs(a["sim"] - b["sim"]) for a, b in zip(got, ref)) <= 2e-4
    metrics = eval_retrieval(got, {"t%d" % i: "v%d" % i for i in range(3)})
    assert set(metrics) == {"text2video", "video2text"} and 0 <= metrics["text2video"]["r1"] <= 100


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16", 3e-2), ("fp16", 4e-3)])
def test_forward_cls_equals_cls_row_of_forward_features(retrieval, mode, tol):
    """TimeSformer.forward_cls (CLS-only tail of the last block, used by the frozen prompter) == forward_features(x)[:, 0]."""
    from alpro_amd import config as rt
    m, batch, _ = retrieval
    x = batch["visual_inputs"].transpose(1, 2)
    with rt.use_compute_dtype(mode), torch.no_grad():
        full = m.visual_encoder.forward_features(x, return_all_tokens=True)[:, 0]
        cls = m.visual_encoder.forward_cls(x)
    assert cls.shape == full.shape
    assert (cls - full).abs().max().item() <= tol * max(1.0, full.abs().max().item())


def test_full_size_batch_properties(bert_cfg):
    """BASELINE-size inputs (64 clips x 8 frames x 224^2, 40-token captions, bf16) have no golden vectors; what must hold at any
    size is checked instead: the encoders are row-independent, so (i) permuting the batch permutes the outputs exactly,
    (ii) one 2B-caption text pass equals two B-caption passes (the batching AlproForPretrain.forward relies on), (iii) the
    forward is idempotent in eval mode, (iv) every output is finite and the VTC similarity rows are proper log-softmax inputs."""
    from alpro_amd import config as rt
    from alpro_amd.modeling.alpro_models import AlproForVideoTextRetrieval
    torch.manual_seed(7)
    m = AlproForVideoTextRetrieval(make_cfg(bert_cfg), dict(VENC, num_frm=8)).eval().cuda()
    B = 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 8, 3, 224, 224, generator=g).cuda()
    ids = torch.randint(1000, 30000, (B, 40), generator=g).cuda()
    ids[:, 0] = 101
    mask = torch.ones(B, 40, dtype=torch.long).cuda()
    mask[::3, 30:] = 0
    perm = torch.randperm(B, generator=g).cuda()
    with rt.use_compute_dtype("bf16"), torch.no_grad():
        ve = m._forward_visual_embeds(x)
        ve_p = m._forward_visual_embeds(x[perm])
        ve_again = m._forward_visual_embeds(x)
        te = m._text_embeds(ids, mask)
        te_cat = m._text_embeds(torch.cat([ids, ids[perm]], 0), torch.cat([mask, mask[perm]], 0))
        out = m(dict(visual_inputs=x, text_input_ids=ids, text_input_mask=mask))
    assert ve.shape == (B, 197, 768) and torch.isfinite(ve).all()
    assert torch.equal(ve_p, ve[perm])                      # (i) exact: a row's arithmetic does not depend on its position
    assert torch.equal(ve_again, ve)                        # (iii)
    assert torch.equal(te_cat[:B], te) and torch.equal(te_cat[B:], te[perm])   # (ii)
    assert out["itm_scores"].shape == (3 * B, 2) and all(torch.isfinite(out[k]).all() for k in ("itm_scores", "itm_loss", "itc_loss"))
    assert 0.0 < float(out["itc_loss"]) < 2 * math.log(B) and 0.0 < float(out["itm_loss"]) < 5.0


# ---- round 2: parity rows that had no GPU test against reference-generated fixtures ------
def _det_block(layer, drop_path):
    from tests.golden.det_init import det_param
    from alpro_amd.modeling.timesformer.vit import Block
    blk = Block(dim=768, num_heads=12, layer_num=layer, mlp_ratio=4.0, qkv_bias=True, drop_path=drop_path, attention_type='divided_space_time')
    with torch.no_grad():
        for k, t in blk.state_dict().items():
            t.copy_(det_param("visual_encoder.model.blocks.%d.%s" % (layer, k), t.shape))
    return blk.cuda()


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-4), ("bf16", 6e-3), ("fp16", 8e-4)])
@pytest.mark.parametrize("path", ["forward_train", "forward"])
def test_block_train_mode_droppath_vs_reference(mode, tol, path):
    """a8: ONE ViT block in TRAIN mode with drop_path 0.1 (vit.py:136-213 + vit_utils.py:137-162) against the reference run with a
    recorded torch.rand stream (tests/golden/block11_droppath_T2_B4.npz): the three Bernoulli row masks are injected into
    DropPath.row_scale's place, everything else is the product path (GEMM-epilogue row scale through the row maps)."""
    from tests.golden.det_init import unit_uniform
    from alpro_amd import config as rt
    g = np.load(os.path.join(GOLDEN, "block11_droppath_T2_B4.npz"))
    B, T, N = 4, 2, 196
    blk = _det_block(11, 0.1).train()
    keep = 1.0 - blk.drop_path.drop_prob
    assert abs(keep - 0.9) < 1e-7
    masks = {rows: (torch.floor(keep + torch.from_numpy(g["rand_%d" % i])) / keep).cuda() for i, rows in enumerate((B * N, B * T, B))}
    assert all(m.numel() == r for r, m in masks.items()) and any(float(m.min()) == 0.0 for m in masks.values())
    blk._drop = lambda rows, device: masks[rows]
    x = torch.from_numpy(unit_uniform("block_in", B * (1 + N * T) * 768).astype(np.float32)).view(B, 1 + N * T, 768).cuda()
    with rt.use_compute_dtype(mode), torch.no_grad():
        y = blk.forward_train(x.clone(), B, T, 14)[0] if path == "forward_train" else blk(x.clone(), B, T, 14)
    scale = float(np.abs(g["y_rows"]).max())
    e = close(y[:, [0, 1, 2, 200, 392]], g["y_rows"], tol * scale, what="block rows (train mode, %s)" % path)
    close(y.norm(dim=-1), g["y_rownorm"], tol * float(g["y_rownorm"].max()), what="block row norms")
    print("\n[droppath block %s/%s] max abs err %.2e (scale %.2f)" % (mode, path, e, scale))


def _sim_ranks(monkeypatch, local_rank, video_feats, text_feats):
    """What tests/golden/ref_harness.set_sim_ranks does to the reference's hvd, applied to alpro_amd.dist: this process plays
    `local_rank` of len(video_feats) ranks; allgather is called for the video features first, then the text features
    (alpro_models.py:110-111)."""
    from alpro_amd import dist
    turn = [0]

    def fake_allgather(x, name=None):
        lst = video_feats if turn[0] % 2 == 0 else text_feats
        turn[0] += 1
        return torch.cat([x if i == local_rank else t.to(x.device) for i, t in enumerate(lst)], 0)
    monkeypatch.setattr(dist, "allgather", fake_allgather)
    monkeypatch.setattr(dist, "local_rank", lambda: local_rank)


def _other_rank_feats(B):
    from tests.golden.det_init import unit_uniform
    ov = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/video", B * 256).astype(np.float32)).view(B, 256), dim=-1)
    ot = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/text", B * 256).astype(np.float32)).view(B, 256), dim=-1)
    return ov, ot


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-3), ("bf16", 1.6e-2), ("fp16", 2e-3)])
def test_world2_vtc_vs_reference(retrieval, monkeypatch, mode, tol):
    """a14 under data parallelism: this process as rank 1 of 2 (the other rank's features are closed-form tensors): VTC targets sit
    at columns [B, 2B) of the gathered similarity and the hard negatives are mined from the rank's own block
    (alpro_models.py:117-123, 289-290), compared with the reference's `w2_*` outputs."""
    from alpro_amd import config as rt
    m, batch, g = retrieval
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    ov, ot = _other_rank_feats(3)
    _sim_ranks(monkeypatch, 1, [ov, None], [ot, None])
    with rt.use_compute_dtype(mode), torch.no_grad():
        out = m(batch)
    for k in ("itc_loss", "itm_loss", "itm_scores"):
        close(out[k], g["w2_" + k], tol, what="w2 " + k)


@pytest.fixture(scope="module")
def prompter(bert_cfg):
    from tests.golden.det_init import det_batch, fill_state_dict_
    from alpro_amd.modeling.alpro_models import Prompter
    m = Prompter(make_cfg(bert_cfg, num_entities=8), dict(VENC, num_frm=2))
    fill_state_dict_(m)
    m.eval().cuda()
    batch = to_dev(det_batch(3, 2, seed_name="prompter_T2", with_mlm=False, with_mpm=True))
    return m, batch, np.load(os.path.join(GOLDEN, "prompter_T2_B3_E8.npz"))


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-3), ("bf16", 3e-2), ("fp16", 3e-3)])
def test_prompter_vs_reference(prompter, monkeypatch, mode, tol):
    """a20 + Prompter.forward: build_text_prompts on 8 entities x 12 (video) / 10 (image) templates, the teacher's VTC forward on
    one rank and as rank 1 of 2, and get_pseudo_labels on both prompt sets, all against the reference
    (alpro_models.py:430-507, 531-551, 553-594)."""
    from tests.golden.det_init import det_prompts
    from alpro_amd import config as rt
    m, batch, g = prompter
    m.prompt_initialized = False
    prompts = dict(batch_enc_video_prompts=det_prompts(8, 12, 15, "prompts/video"), batch_enc_image_prompts=det_prompts(8, 10, 15, "prompts/image"))
    e = {}
    with rt.use_compute_dtype(mode), torch.no_grad():
        m.build_text_prompts(prompts)
        assert m.prompt_initialized and m.video_prompt_feat.shape == (8, 256)
        e["video_prompt_feat"] = close(m.video_prompt_feat, g["video_prompt_feat"], tol * 0.1, what="video_prompt_feat")
        e["image_prompt_feat"] = close(m.image_prompt_feat, g["image_prompt_feat"], tol * 0.1, what="image_prompt_feat")
        out = m(batch)
        for k in ("itc_loss", "i2t_scores", "t2i_scores"):
            e[k] = close(out[k], g[k], tol, what=k)
        assert torch.equal(out["itc_labels"].cpu(), torch.from_numpy(g["itc_labels"]).long())
        for ty in ("video", "img"):
            soft, ign = m.get_pseudo_labels(dict(batch, type=ty))
            e["pseudo_" + ty] = close(soft, g["pseudo_labels_" + ty], tol, what="pseudo labels " + ty)
            assert torch.equal(ign.cpu().float(), torch.from_numpy(g["pseudo_ignore_" + ty]))
        ov, ot = _other_rank_feats(3)
        _sim_ranks(monkeypatch, 1, [ov, None], [ot, None])
        out2 = m(batch)
        for k in ("itc_loss", "i2t_scores", "t2i_scores"):
            close(out2[k], g["w2_" + k], tol, what="w2 " + k)
        assert torch.equal(out2["itc_labels"].cpu(), torch.from_numpy(g["w2_itc_labels"]).long())
    with pytest.raises(AssertionError, match="Repetitively"):
        m.build_text_prompts(prompts)
    print("\n[prompter parity %s]" % mode, {k: "%.2e" % v for k, v in e.items()})


@pytest.mark.parametrize("mode,rtol", [("fp32", 5e-3), ("bf16", 4e-2), ("fp16", 1e-2)])
def test_retrieval_finetune_gradients_vs_reference(fixture_models, monkeypatch, mode, rtol):
    """BASELINE configs[4] (retrieval finetune step): loss = itm_loss + itc_loss (run_video_retrieval.py:432-434) backward through
    AlproForVideoTextRetrieval on the HIP backward; gradient norms of every trained tensor + 12 full gradients vs the reference."""
    from alpro_amd import config as rt
    g = np.load(os.path.join(GOLDEN, "retrieval_grads_T2_B3.npz"))
    m, batch, _ = fixture_models("retrieval_T2")
    fresh_grads(m)
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    with rt.use_compute_dtype(mode):
        out = m(batch)
        gs = backward(out["itm_loss"] + out["itc_loss"], mode)
    tol = {"fp32": 1e-3, "fp16": 2e-3, "bf16": 1.6e-2}[mode]
    for k in ("itc_loss", "itm_loss", "itm_scores"):
        close(out[k], g[k], tol, what=k + " (train graph)")
    pd = dict(m.named_parameters())
    if gs != 1.0:
        for p_ in pd.values():
            if p_.grad is not None:
                p_.grad.div_(gs)
    names = [str(n) for n in g["grad_norm_names"]]
    missing = [n for n in names if pd[n].grad is None]
    assert not missing, "no gradient for %s" % missing[:5]
    extra = [n for n, p in pd.items() if p.grad is not None and n not in names]
    assert not extra, "unexpected gradients %s" % extra[:5]
    got = np.array([float(pd[n].grad.norm()) for n in names])
    ref = g["grad_norms"]
    # `temp` is ONE scalar whose gradient is a cancelling sum over the similarity matrix (0.04 on the released-geometry fixture against ~2
    # on retrieval_T2): its error is measured against the scale of the non-cancelling case, not against its own near-zero value
    rel = np.abs(got - ref) / np.where(np.array([n == "temp" for n in names]), np.maximum(ref, 0.5), np.maximum(ref, 1e-5))
    zero_grad = np.array([n.endswith("attention.self.key.bias") for n in names])   # exactly 0 in exact arithmetic (see the pretrain test)
    assert got[zero_grad].max() < 1e-3
    rel[zero_grad] = 0.0
    worst = int(rel.argmax())
    print("\n[retrieval grad parity %s] worst grad-norm rel err %.2e at %s; median %.2e" % (mode, rel.max(), names[worst], np.median(rel)))
    if os.environ.get("ALPRO_PARITY_REPORT"):
        print("[parity-report] retrieval_gradients[%s] | worst grad-norm rel err | err %.3e | limit %.1e" % (mode, rel.max(), rtol))
        return
    assert rel.max() < rtol, (names[worst], got[worst], ref[worst])
    for k in g.files:
        if k.startswith("grad/"):
            r = g[k].astype(np.float64)
            e = np.abs(pd[k[5:]].grad.float().cpu().numpy().astype(np.float64) - r).max()
            assert e <= rtol * max(np.abs(r).max(), 1e-6) + 1e-7, (k, e, np.abs(r).max())


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-3), ("bf16", 3e-2), ("fp16", 3e-3)])
def test_retrieval_16_frames_vs_reference(fixture_models, monkeypatch, mode, tol):
    """Model-level case at 16 frames per clip (BASELINE configs[4]): forward, visual embeddings, 1-video-x-n-captions inference."""
    from alpro_amd import config as rt
    g = np.load(os.path.join(GOLDEN, "retrieval_T16_B2.npz"))
    m, batch, _ = fixture_models("retrieval_T16")
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    with rt.use_compute_dtype(mode), torch.no_grad():
        out = m(batch)
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        inf = m.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1], text_input_ids=batch["text_input_ids"],
                                       text_input_mask=batch["text_input_mask"]))
    for k in ("itc_loss", "itm_loss", "itm_scores"):
        close(out[k], g[k], tol, what=k)
    assert torch.equal(out["itm_labels"].cpu(), torch.from_numpy(g["itm_labels"]).long())
    close(ve[:, [0, 1, 100, 196]], g["video_embeds_rows"], tol * (1 if mode == "fp32" else 2), what="video_embeds rows (16 frames)")
    close(inf["itc_scores"], g["inf_itc_scores"], {"fp32": 1e-3, "fp16": 1e-3}.get(mode, tol), what="VTC logits (16 frames): the north star's 1e-3 for fp32 and fp16")
    close(inf["logits"], g["inf_logits"], tol, what="inference ITM logits (16 frames)")


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-4), ("bf16", 8e-3), ("fp16", 1e-3)])
def test_batched_retrieval_scoring_vs_reference_records(fixture_models, mode, tol):
    """N3: every caption against every cached video in flat fusion mini-batches (score_all_pairs) reproduces the records the
    REFERENCE's evaluation loop produced for 5 videos x 5 captions (tests/golden/retrieval_eval_T2_V5.npz: forward_inference per
    (video, 3-caption mini-batch), softmax of the ITM logits, ITC similarity, both rounded to 4 decimals), and the cached loop
    (inference_retrieval_cached) gives the same numbers."""
    from tests.golden.det_init import det_batch
    from alpro_amd import config as rt
    from alpro_amd.retrieval_eval import inference_retrieval_cached, records_from_matrices, retrieval_metrics_on_device, score_all_pairs
    g = np.load(os.path.join(GOLDEN, "retrieval_eval_T2_V5.npz"))
    V = 5
    m = fixture_models("retrieval_T2")[0]     # the same 2-frame retrieval model (closed-form weights), its own 5 videos x 5 captions
    batch = to_dev(det_batch(V, 2, seed_name="retrieval_eval_T2", with_mlm=False, with_mpm=False))
    with rt.use_compute_dtype(mode):
        score, sim = score_all_pairs(m, batch["visual_inputs"], batch["text_input_ids"], batch["text_input_mask"], pair_bsz=7)
        recs = inference_retrieval_cached(m, [("v%d" % i, batch["visual_inputs"][i:i + 1]) for i in range(V)], batch["text_input_ids"],
                                          batch["text_input_mask"], ["t%d" % i for i in range(V)], eval_bsz=3)
    assert score.is_cuda and score.shape == (V, V)
    close(score, g["score"], tol + 5e-5, what="ITM match probability (V x C)")       # the fixture is rounded to 4 decimals
    close(sim, g["sim"], tol * 5 + 5e-5, what="ITC similarity (V x C)")
    mine = records_from_matrices(score, sim, ["v%d" % i for i in range(V)], ["t%d" % i for i in range(V)])
    assert [(r["vid_id"], r["txt_id"]) for r in mine] == [(r["vid_id"], r["txt_id"]) for r in recs]
    assert max(abs(a["score"] - b["score"]) for a, b in zip(mine, recs)) <= 2e-4 + tol
    dm = retrieval_metrics_on_device(score, torch.arange(V))
    assert 0 <= dm["r1"] <= 100 and 1 <= dm["medianR"] <= V


@pytest.mark.parametrize("mode,tol,rtol", [("fp32", 1e-3, 5e-3), ("fp16", 4e-3, 6e-3), ("bf16", 3e-2, 4e-2)])
def test_pretrain_released_geometry_vs_reference(fixture_models, monkeypatch, mode, tol, rtol):
    """the geometry the reference actually pretrains with (config_release/pretrain_alpro.json:34,37,59 -- 4 frames,
    30-token captions, fusion sequences of 227 tokens): every loss, the VTC logits, ITM scores, MLM columns, embeddings and the
    parameter-gradient norms of loss = mlm + itm + itc + mpm against the reference-generated fixture."""
    from alpro_amd import config as rt
    g = np.load(os.path.join(GOLDEN, "pretrain_release_T4_L30_B2.npz"))
    m, batch, _ = fixture_models("pretrain_release_T4_L30")
    fresh_grads(m)
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    with rt.use_compute_dtype(mode):
        with torch.no_grad():
            ve = m._forward_visual_embeds(batch["visual_inputs"])
            te, tf = m._forward_text_feats(batch)
            vf = m._video_feat(ve)
        keep = arm_scale(mode)
        out = m(batch)
        gs = backward(out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"], mode)
        del keep
    assert out["mlm_scores"].shape == (2, 30, 30522)
    for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "mpm_logits"):
        close(out[k], g[k], tol, what=k)
    close(out["mlm_scores"][:, :, ::61], g["mlm_scores_cols"], tol, what="mlm_scores")
    close(vf @ tf.t() / m.temp, g["sim_v2t"], {"fp32": 1e-3, "fp16": 1e-3, "bf16": 1.6e-2}[mode],
          what="VTC logits (released geometry): the north star's 1e-3 for fp32 and fp16 (fp16 = 16-bit operands + precise CLS rows since round 4)")
    close(te[:, [0, 1, 29]], g["text_embeds_rows"], tol * (1 if mode == "fp32" else 2), what="text_embeds rows")
    close(ve[:, [0, 1, 57, 196]], g["video_embeds_rows"], tol * (1 if mode == "fp32" else 2), what="video_embeds rows")
    assert torch.equal(out["itm_labels"].cpu(), torch.from_numpy(g["itm_labels"]).long())
    pd = dict(m.named_parameters())
    names = [str(n) for n in g["grad_norm_names"]]
    assert not [n for n in names if pd[n].grad is None]
    got = np.array([float(pd[n].grad.norm()) / gs for n in names])
    ref = g["grad_norms"]
    # `temp` is ONE scalar whose gradient is a cancelling sum over the similarity matrix (0.04 on the released-geometry fixture against ~2
    # on retrieval_T2): its error is measured against the scale of the non-cancelling case, not against its own near-zero value
    rel = np.abs(got - ref) / np.where(np.array([n == "temp" for n in names]), np.maximum(ref, 0.5), np.maximum(ref, 1e-5))
    rel[np.array([n.endswith("attention.self.key.bias") for n in names])] = 0.0
    print("\n[released-geometry grad parity %s] worst grad-norm rel err %.2e at %s; median %.2e" % (mode, rel.max(), names[int(rel.argmax())], np.median(rel)))
    if os.environ.get("ALPRO_PARITY_REPORT"):
        print("[parity-report] released_geometry_gradients[%s] | worst grad-norm rel err | err %.3e | limit %.1e" % (mode, rel.max(), rtol))
        return
    assert rel.max() < rtol, (names[int(rel.argmax())], float(rel.max()))

# ---- round 4: the bar on the worst fixture, and parity at the benchmarked size ------------------------------------
MODES = {"fp32": ("fp32", "auto"), "fp16": ("fp16", "auto"), "fp16_plain": ("fp16", "0"), "bf16": ("bf16", "auto"), "bf16_cls": ("bf16", "1")}


@pytest.mark.parametrize("case", ["retrieval_T2", "pretrain_T8", "retrieval_T16", "pretrain_release_T4_L30"])
def test_vtc_logits_meet_the_north_star_bar_on_every_fixture(fixture_models, case):
    """BASELINE.json: "VTC logits within 1e-3 of reference" -- on EVERY reference-generated fixture, in the exact mode and in the benchmark's
    mode (fp16 operands + precise CLS rows).  Plain fp16 (no CLS side path: 3.4e-4 .. 1.06e-3 in round 3) and bf16 are measured beside
    them with the limits of what they reach."""
    from alpro_amd import config as rt
    from tests.golden import parity_cases as pc
    m, batch, ref = fixture_models(case)
    errs = {}
    for name, (dt, cls) in MODES.items():
        with rt.use_compute_dtype(dt), rt.use_cls_precise(cls):
            errs[name] = pc.vtc_logit_error(case, m, batch, ref)
    print("\n[vtc-logit parity %s] " % case + "  ".join("%s %.2e" % kv for kv in errs.items()))
    if os.environ.get("ALPRO_PARITY_REPORT"):
        for k, v in errs.items():
            print("[parity-report] vtc_logits[%s] | %s | err %.3e | limit %.1e" % (case, k, v, pc.NORTH_STAR_BAR if k in ("fp32", "fp16") else 2e-2))
        return
    assert errs["fp32"] <= 2e-5, errs
    assert errs["fp16"] <= pc.NORTH_STAR_BAR, errs
    assert errs["fp16_plain"] <= 2e-3 and errs["bf16"] <= 1.6e-2 and errs["bf16_cls"] <= 1.6e-2, errs


def test_full_size_pretrain_forward_in_the_bench_dtype_vs_the_exact_mode(bert_cfg, monkeypatch):
    """No golden vectors exist at BASELINE's size, but the exact fp32 HIP mode is pinned to the reference at <= 5e-6 on every fixture: it is
    the oracle at full size.  AlproForPretrain, eval mode, B = 64 x 8 frames x 224^2 + 40 tokens, random-init weights: the benchmark's mode
    (fp16 operands + precise CLS rows) against the exact mode on ALL 4096 VTC logits (max and p99.9), the ITM scores of the same hard
    negatives, the MPM logits and every 61st MLM column."""
    from alpro_amd import config as rt
    from alpro_amd.modeling.alpro_models import AlproForPretrain
    import bench
    torch.manual_seed(4)
    B, T = 64, 8
    m = AlproForPretrain(make_cfg(bert_cfg), dict(VENC, num_frm=T)).eval().cuda()
    batch = bench.synth_batch(B, T, "cuda", seed=11, full=True)
    batch["text_input_mask"] = batch["text_input_mask"].clone()
    batch["text_input_mask"][::3, 31:] = 0
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    negs = {}
    orig = AlproForPretrain._sample_negatives

    def record(sim_v2t, sim_t2v, bs):
        if "n" not in negs:
            negs["n"] = orig(sim_v2t, sim_t2v, bs)
        return negs["n"]
    monkeypatch.setattr(AlproForPretrain, "_sample_negatives", staticmethod(record))

    def run(dt, cls):
        with rt.use_compute_dtype(dt), rt.use_cls_precise(cls), torch.no_grad():
            ve = m._forward_visual_embeds(batch["visual_inputs"])
            _, tf = m._forward_text_feats(batch)
            logits = m._video_feat(ve) @ tf.t() / m.temp
            out = m(batch)
        return dict(logits=logits.double().cpu(), itm=out["itm_scores"].double().cpu(), mpm=out["mpm_logits"].double().cpu(),
                    mlm=out["mlm_scores"][:, :, ::61].double().cpu(), itc_loss=out["itc_loss"].double().cpu())
    ref = run("fp32", "auto")        # (runs first: its hard negatives are the ones every mode scores)
    rep = {}
    for name, (dt, cls) in MODES.items():
        if name == "fp32":
            continue
        got = run(dt, cls)
        e = (got["logits"] - ref["logits"]).abs().flatten()
        rep[name] = dict(max=float(e.max()), p999=float(e.kthvalue(int(0.999 * e.numel()))[0]), rms=float(e.pow(2).mean().sqrt()),
                         itm=float((got["itm"] - ref["itm"]).abs().max()), mpm=float((got["mpm"] - ref["mpm"]).abs().max()),
                         mlm=float((got["mlm"] - ref["mlm"]).abs().max()), itc_loss=float((got["itc_loss"] - ref["itc_loss"]).abs()))
        print("\n[B=64 proxy %-10s] VTC logits (4096): max %.2e p99.9 %.2e rms %.2e | ITM %.2e | MPM %.2e | MLM %.2e | itc_loss %.2e" % (
            name, rep[name]["max"], rep[name]["p999"], rep[name]["rms"], rep[name]["itm"], rep[name]["mpm"], rep[name]["mlm"], rep[name]["itc_loss"]))
        if os.environ.get("ALPRO_PARITY_REPORT"):
            print("[parity-report] full_size_proxy[%s] | VTC logits max / p99.9 / rms | err %.3e / %.3e / %.3e | limit 1.0e-03" % (name, rep[name]["max"], rep[name]["p999"], rep[name]["rms"]))
    if os.environ.get("ALPRO_PARITY_REPORT"):
        return
    # round 6: the north star's 1e-3 on the MAXIMUM of the 4096 logits, not only on p99.9 (measured: max 5.4e-4, p99.9 5.0e-4,
    # profiles/r4_parity_pareto.txt; rounds 4-5 allowed 1.5e-3 on the maximum)
    assert rep["fp16"]["max"] <= 1e-3 and rep["fp16"]["p999"] <= 1e-3, rep["fp16"]
    assert rep["fp16"]["rms"] < 0.6 * rep["fp16_plain"]["rms"], rep                        # the side path must carry its weight at full size too
    assert rep["fp16"]["itm"] <= 5e-3 and rep["fp16"]["mlm"] <= 2e-2 and rep["fp16"]["mpm"] <= 5e-3, rep["fp16"]


def test_full_size_pretrain_backward_in_the_bench_dtype_vs_the_exact_mode(bert_cfg):
    """the hand-written backward at the benchmarked size.  AlproForPretrain, B = 64 x 8 frames x 224^2 + 40 tokens, one
    forward + backward of mlm + itm + itc + mpm (run_pretrain_sparse.py:557,595-601) with drop-path / dropout at 0: the benchmark's mode (fp16
    operands + precise CLS rows, loss scale 2^16) against the exact fp32 HIP mode -- the oracle at this size, itself held to the reference's
    gradients on the fixtures above -- for EVERY parameter tensor that receives a gradient: relative error of the norm, cosine, relative L2
    error; plus the four losses.  The bounds are what the first measurement showed, with margin (tests/golden/parity_cases.py)."""
    from tests.golden import parity_cases as pc
    rep = pc.full_size_backward_parity(bert_cfg, VENC, make_cfg, "cuda")
    print("\n[B=64 backward, %s vs exact fp32] %d tensors | norm rel err worst %.2e median %.2e | cosine worst %.6f median %.6f | l2 rel err worst %.2e median %.2e | "
          "global norm %.2e cosine %.6f | losses %s | worst %s" % (rep["mode"], rep["grad_tensors"], rep["grad_norm_rel_err_worst"], rep["grad_norm_rel_err_median"],
                                                                   rep["grad_cosine_worst"], rep["grad_cosine_median"], rep["grad_l2_rel_err_worst"], rep["grad_l2_rel_err_median"],
                                                                   rep["global_grad_norm_rel_err"], rep["global_grad_cosine"],
                                                                   {k: float("%.2e" % v) for k, v in rep["loss_abs_err"].items()}, rep["worst_tensors"]))
    assert not rep["nonfinite_tensors"], rep["nonfinite_tensors"]
    assert rep["grad_tensors"] >= 440, rep["grad_tensors"]
    lim = pc.FULL_SIZE_BACKWARD_LIMITS
    assert rep["grad_l2_rel_err_worst"] <= lim["grad_l2_rel_err_worst"] and rep["grad_l2_rel_err_median"] <= lim["grad_l2_rel_err_median"], rep
    assert rep["grad_norm_rel_err_worst"] <= lim["grad_norm_rel_err_worst"] and rep["grad_norm_rel_err_median"] <= lim["grad_norm_rel_err_median"], rep
    assert rep["grad_cosine_worst"] >= lim["grad_cosine_worst"] and rep["grad_cosine_median"] >= lim["grad_cosine_median"], rep
    assert rep["global_grad_cosine"] >= lim["global_grad_cosine"] and rep["global_grad_norm_rel_err"] <= lim["global_grad_norm_rel_err"], rep
    assert all(v <= lim["loss_abs_err"] for v in rep["loss_abs_err"].values()), rep["loss_abs_err"]


def test_finetune_backward_at_the_msrvtt_geometry_in_the_bench_dtype_vs_the_exact_mode(bert_cfg):
    """BASELINE configs[4]'s finetune step at config_release/msrvtt_ret.json's own geometry (num_frm 8, train_batch_size 8,
    max_txt_len 40) -- the reference-generated gradient fixture for this model is T = 2, B = 3 (retrieval_grads_T2_B3.npz).  AlproForVideoTextRetrieval,
    loss = itm_loss + itc_loss (run_video_retrieval.py:432-434), one forward + backward in the benchmark's mode (fp16 operands + precise CLS rows,
    loss scale 2^16) against the exact fp32 HIP mode on the same weights, inputs and hard negatives, every parameter gradient.  Same bounds as
    the B = 64 pretraining step (tests/golden/parity_cases.py::FULL_SIZE_BACKWARD_LIMITS)."""
    from tests.golden import parity_cases as pc
    rep = pc.full_size_backward_parity(bert_cfg, VENC, make_cfg, "cuda", B=8, T=8, model="retrieval")
    print("\n[msrvtt_ret geometry backward, %s vs exact fp32] %d tensors | l2 rel err worst %.2e median %.2e | cosine worst %.6f | global norm %.2e cosine %.6f | losses %s | worst %s" % (
        rep["mode"], rep["grad_tensors"], rep["grad_l2_rel_err_worst"], rep["grad_l2_rel_err_median"], rep["grad_cosine_worst"], rep["global_grad_norm_rel_err"],
        rep["global_grad_cosine"], {k: float("%.2e" % v) for k, v in rep["loss_abs_err"].items()}, rep["worst_tensors"]))
    assert not rep["nonfinite_tensors"], rep["nonfinite_tensors"]
    assert rep["grad_tensors"] >= 400, rep["grad_tensors"]
    lim = pc.FULL_SIZE_BACKWARD_LIMITS
    assert rep["grad_l2_rel_err_worst"] <= lim["grad_l2_rel_err_worst"] and rep["grad_l2_rel_err_median"] <= lim["grad_l2_rel_err_median"], rep
    assert rep["grad_cosine_worst"] >= lim["grad_cosine_worst"] and rep["global_grad_cosine"] >= lim["global_grad_cosine"], rep
    # the whole-gradient norm at 8 pairs: first measurement 5.4e-4 (439 tensors: l2 worst 2.8e-3 / median 9.2e-4, cosine worst 0.999996, losses <= 9.5e-5;
    # gpurun_out r6c1) -- a sum over 8 x 1569 tokens instead of 64 x 1569 averages less rounding noise away than the B = 64 step's 7.1e-5 (limit 5e-4 there)
    assert rep["global_grad_norm_rel_err"] <= 1.5e-3, rep
    assert all(v <= lim["loss_abs_err"] for v in rep["loss_abs_err"].values()), rep["loss_abs_err"]


@pytest.mark.parametrize("world,rank", [(1, 0), (2, 1)])
def test_hard_negative_sampler_unpatched(monkeypatch, world, rank):
    """every VTM parity test above replaces torch.multinomial by argmax.  This one runs the product's batched sampler
    (alpro_models.py::_sample_negatives: ONE torch.multinomial per direction on the device) as it is and holds it to what the reference's per-row
    loop does (alpro_models.py:287-313): the candidates are the rank's OWN block of the gathered similarity (columns [b*rank, b*(rank+1)) --
    simulated world 2, rank 1), the positive (diagonal) is never drawn, and over 4000 draws the empirical frequency of every candidate is within
    4 sigma of the softmax weight of its similarity (binomial standard error; 4 sigma over 2 x 16 x 15 cells: false-alarm rate < 2 %)."""
    from tests.golden import parity_cases as pc
    pc.sampler_property_check("cuda", world, rank, 4000, monkeypatch)


def test_cls_chain_on_a_side_stream_is_result_neutral():
    """Round 5 (alpro_amd.config.cls_stream): the precise-CLS chain of the ViT blocks issued on a second HIP stream -- the same launches, ordered
    against the main path by events instead of by stream order -- gives bit-identical results on every path that carries it: the in-place
    inference forward (also through forward_cls, the prompter's entry), and the out-of-place training forward + hand-written backward
    (outputs, saved pre-MLP CLS rows via the parameter gradients).  Several passes back to back: the side buffers are re-used from block to
    block and from pass to pass."""
    from alpro_amd import config as rt
    from alpro_amd.modeling.timesformer.vit import TimeSformer
    torch.manual_seed(21)
    T, B = 4, 6
    enc = TimeSformer(dict(VENC, num_frm=T, drop_path_rate=0.0), input_format="RGB").cuda()
    x = torch.randn(B, 3, T, 224, 224, device="cuda")
    dout = torch.randn(B, 197, 768, device="cuda") * 1e-2
    res = {}
    prev = rt._cls_stream[0]
    try:
        for on in (False, True, True, False):
            rt.set_cls_stream(on)
            with rt.use_compute_dtype("bf16"), rt.use_cls_precise("1"):
                enc.eval()
                with torch.no_grad():
                    y = enc.forward_features(x)
                    c = enc.forward_cls(x)
                enc.train()
                for p in enc.parameters():
                    p.grad = None
                with torch.enable_grad():
                    yt = enc.forward_features(x)
                    (yt * dout).sum().backward()
                g = torch.cat([p.grad.reshape(-1) for p in enc.parameters() if p.grad is not None])
            torch.cuda.synchronize()
            cur = (y.clone(), c.clone(), yt.detach().clone(), g.clone())
            if not res:
                res["ref"] = cur
            else:
                for a, b, what in zip(cur, res["ref"], ("forward_features", "forward_cls", "training forward", "parameter gradients")):
                    assert torch.equal(a, b), "%s differs with cls_stream=%s" % (what, on)
    finally:
        rt.set_cls_stream(prev)
    assert bool(torch.isfinite(res["ref"][3]).all()) and float(res["ref"][3].abs().sum()) > 0


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_fused_temporal_half_against_the_two_launches(mode):
    """Round 6 (alpro_amd.config.fuse_temporal_attention; csrc/gemm_tattn.hip): the temporal half's qkv Linear + frame attention as one launch, on
    every path that carries it -- the in-place inference forward, forward_cls (the prompter's entry) and, opt-in, the training forward with its
    hand-written backward (the kernel then also writes q | k | v and the log-sum-exp rows) -- against the two launches on the same weights and
    clips.  Same roundings on both sides (q, k, v, P and the attention output to 16 bits), different fp32 summation orders inside the softmax:
    the encoder outputs agree to a few units of the operand dtype's resolution, the parameter gradients to 1e-3 of their norm."""
    from alpro_amd import config as rt
    from alpro_amd.modeling.timesformer.vit import TimeSformer
    torch.manual_seed(31)
    T, B = 4, 4                                  # B * 196 * T = 3136 rows: 12 full 256-row panels + a ragged one of 64 rows
    enc = TimeSformer(dict(VENC, num_frm=T, drop_path_rate=0.0), input_format="RGB").cuda()
    with torch.no_grad():                        # TimeSformer zero-initialises temporal_fc of blocks 1..11: the temporal halves would not reach the output
        for blk in enc.model.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    x = torch.randn(B, 3, T, 224, 224, device="cuda")
    dout = torch.randn(B, 197, 768, device="cuda") * 1e-2
    prev = rt._fuse_tattn[0]
    res = {}
    try:
        for fuse in ("0", "1"):
            rt.set_fuse_temporal_attention(fuse)
            with rt.use_compute_dtype(mode), rt.use_cls_precise("0"):
                enc.eval()
                with torch.no_grad():
                    y = enc.forward_features(x).float().clone()
                    c = enc.forward_cls(x).float().clone()
                enc.train()
                for p in enc.parameters():
                    p.grad = None
                sc = arm_scale(mode)
                with torch.enable_grad():
                    yt = enc.forward_features(x)
                    gs = backward((yt * dout).sum(), mode)
                g = torch.cat([p.grad.reshape(-1) for p in enc.parameters() if p.grad is not None]).double() / gs
                del sc
            torch.cuda.synchronize()
            res[fuse] = (y, c, yt.detach().float().clone(), g)
    finally:
        rt.set_fuse_temporal_attention(prev)
        rt.set_armed_loss_scaler(None)
    eps = {"fp16": 2.0 ** -10, "bf16": 2.0 ** -7}[mode]
    for i, what in enumerate(("forward_features", "forward_cls", "training forward")):
        a, b = res["1"][i], res["0"][i]
        err = float((a - b).abs().max())
        assert err <= 8 * eps * max(1.0, float(b.abs().max())), "%s: fused vs two launches %.3e" % (what, err)
    ga, gb = res["1"][3], res["0"][3]
    rel = float((ga - gb).norm() / gb.norm())
    print("\n[fused temporal half %s] outputs within 8 eps; gradient l2 rel diff %.2e" % (mode, rel))
    assert bool(torch.isfinite(ga).all()) and rel <= {"fp16": 1e-3, "bf16": 8e-3}[mode], rel


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("fp16", 4e-3)])
def test_last_fusion_layer_tail_on_the_read_rows_only(fixture_models, monkeypatch, mode, tol):
    """Round 6 (AlproForPretrain.fusion_tail_rows; BertLayer.forward_train rows=...): the last fusion layer's attention-output dense, LayerNorms and FFN
    run on the 239 of every 948 rows the heads read ([CLS] rows, MLM text rows, the positives' patch rows) -- dead-row elimination, forward and
    backward.  Every output of the pretraining forward and every parameter gradient against the every-row form on the 8-frame fixture model: the same
    values up to the summation order of GEMMs over fewer rows (fp32: 2e-5 of the tensor's scale; fp16 operands: a few units of their resolution)."""
    from alpro_amd import config as rt
    m, batch, _ = fixture_models("pretrain_T8")
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    monkeypatch.setattr(np.random, "uniform", lambda *a, **k: 1.0)
    prev, was_training = m.fusion_tail_rows, m.training
    res = {}
    try:
        for tail in (False, True):
            m.fusion_tail_rows = tail
            fresh_grads(m)
            with rt.use_compute_dtype(mode), torch.enable_grad():
                sc = arm_scale(mode)
                out = m(batch)
                loss = out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]
                gs = backward(loss, mode)
                del sc
            res[tail] = ({k: out[k].detach().float().clone() for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "mpm_logits", "mlm_scores")},
                         {n: (p.grad.detach().double() / gs).clone() for n, p in m.named_parameters() if p.grad is not None})
            torch.cuda.synchronize()
    finally:
        m.fusion_tail_rows = prev
        rt.set_armed_loss_scaler(None)
        fresh_grads(m)
        m.train(was_training)
    (oa, ga), (ob, gb) = res[False], res[True]
    for k in oa:
        err = float((oa[k] - ob[k]).abs().max())
        assert err <= tol * max(1.0, float(oa[k].abs().max())), "%s: %.3e" % (k, err)
    assert set(ga) == set(gb)
    worst = max(((float((ga[n] - gb[n]).norm() / ga[n].norm().clamp_min(1e-12)), n) for n in ga if float(ga[n].norm()) > 0), default=(0.0, ""))
    print("\n[fusion tail rows %s] worst relative gradient difference %.2e (%s)" % (mode, worst[0], worst[1]))
    assert worst[0] <= (1e-4 if mode == "fp32" else 2e-2), worst


def test_side_stream_backward_is_handed_to_the_callers_stream():
    """The mechanisms behind ALPRO_TEXT_STREAM and ALPRO_WGRAD_STREAM, isolated.
    (a) An anchored run whose forward was queued on the text side stream gets its backward on that stream (autograd's rule), where it writes a buffer
        behind autograd's back -- as the hand-written backwards write parameter gradients -- after a long spin (torch.cuda._sleep: ~0.3 s of device
        time, so an unordered read on the caller's stream WOULD see the old value).  When loss.backward() returns, the caller's stream must be ordered
        behind that write (alpro_amd.modeling.train.Anchor.backward queues an engine callback that makes the caller's stream wait for an event of the
        side stream).  A backward on the text side stream keeps its weight gradients there (no side stream of a side stream: profiles/r6_hw_queues.txt).
    (b) An anchored run on the launch stream sends a "weight gradient" to the weight-gradient side stream from inside its backward (what
        alpro_amd.modeling.train.wgrad does); the launch stream must be ordered behind it where the backward returns.
    Negative controls run once by hand on torch 2.10 (not part of the suite): without `join_wgrad` (b) fails, as it must; without the callback (a)
    still holds -- this torch's engine orders the caller's stream behind every stream a backward node ran on by itself (a plain autograd.Function on
    a side stream shows the same) -- so the callback is the documented guarantee for engines that do not, and this test pins the property either way.
    The streams are picked among eight candidates that demonstrably run beside the launch stream."""
    from alpro_amd import config as rt
    from alpro_amd.modeling import train as tr
    dev = torch.device("cuda", torch.cuda.current_device())
    prev_t, prev_w = rt.text_stream_enabled(), rt.wgrad_stream_enabled()
    rt.set_text_stream(True)
    rt.set_wgrad_stream(True)
    seen = {}

    class Run:
        def __init__(self, tag):
            self.tag = tag

        def forward(self, x):
            return x * 2.0

        def backward(self, g):
            seen[self.tag + "_stream"] = torch.cuda.current_stream().cuda_stream
            w = rt.wgrad_side_stream(g.device)          # what tr.wgrad asks for a weight-gradient GEMM
            seen[self.tag + "_wgrad_side"] = w
            if w is None:                                # (a): everything on the stream of the backward
                torch.cuda._sleep(600_000_000)
                self.buf.fill_(7.0)
            else:                                        # (b): behind the launch stream, on its side stream
                w.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(w):
                    torch.cuda._sleep(600_000_000)
                    self.buf.fill_(9.0)
            return g * 2.0

    def runs_beside(main, cand):
        """Does `cand` execute independently of `main`?  (HIP maps streams onto a few hardware queues; two streams that share one run in submission
        order, which would order the caller's stream behind the side stream by accident and make this test prove nothing.)"""
        flag = torch.zeros(1, device=dev)
        torch.cuda.synchronize()
        cand.wait_stream(main)
        with torch.cuda.stream(cand):
            torch.cuda._sleep(300_000_000)
            flag.fill_(1.0)
        early = flag.clone()          # on main, unordered: sees 0 if the candidate's spin has not held it up
        torch.cuda.synchronize()
        return float(early.item()) == 0.0
    main = torch.cuda.current_stream(dev)
    key = (dev.index, main.cuda_stream)
    prev_side, prev_wside = rt._TEXT_SIDE.get(key), rt._WGRAD_SIDE.copy()
    try:
        cands = [torch.cuda.Stream(dev) for _ in range(8)]
        free = [c for c in cands if runs_beside(main, c)]
        assert len(free) >= 2, "no stream of eight runs beside the launch stream on this box"
        rt._TEXT_SIDE[key] = free[0]                 # the text side stream of this launch stream ...
        rt._WGRAD_SIDE[key] = [free[1], False]       # ... and its weight-gradient side stream
        side = rt.text_side_stream(dev)
        assert side is free[0] and side.cuda_stream != main.cuda_stream
        # no gradient flows OUT of the runs to a tensor of the caller's stream (as with the text encoder: token ids in, parameters only) -- otherwise the
        # engine's own producer -> consumer ordering for that gradient would hide a missing hand-over
        x = torch.ones(8, device=dev)
        p = torch.nn.Parameter(torch.zeros(1, device=dev))
        # (a)
        a = Run("a")
        a.buf = torch.zeros(4, device=dev)
        torch.cuda.synchronize()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            y = tr.run_anchored(a, [x], [p])
        main.wait_stream(side)
        y.record_stream(main)
        assert y.requires_grad
        y.sum().backward()
        got_a = a.buf.clone()     # queued on the caller's stream right behind backward(): no device-wide sync in between
        torch.cuda.synchronize()
        assert seen["a_stream"] == side.cuda_stream, "autograd did not run the anchored backward on the stream of its forward"
        assert seen["a_wgrad_side"] is None, "a backward on the text side stream was given a side stream of its own"
        assert got_a.tolist() == [7.0] * 4, "the caller's stream read the side stream's write too early: %s" % got_a.tolist()
        # (b)
        b = Run("b")
        b.buf = torch.zeros(4, device=dev)
        torch.cuda.synchronize()
        y = tr.run_anchored(b, [x], [p])
        y.sum().backward()
        got_b = b.buf.clone()
        torch.cuda.synchronize()
        assert seen["b_stream"] == main.cuda_stream and seen["b_wgrad_side"] is free[1], "no weight-gradient side stream inside an anchored backward on the launch stream"
        assert got_b.tolist() == [9.0] * 4, "the weight-gradient side stream was not joined where the backward returned: %s" % got_b.tolist()
    finally:
        torch.cuda.synchronize()
        rt.set_text_stream(prev_t)
        rt.set_wgrad_stream(prev_w)
        if prev_side is None:
            rt._TEXT_SIDE.pop(key, None)
        else:
            rt._TEXT_SIDE[key] = prev_side
        rt._WGRAD_SIDE.clear()
        rt._WGRAD_SIDE.update(prev_wside)


def test_pretrain_step_never_makes_the_host_wait_for_the_device(fixture_models):
    """Round 6 (profiles/r6_host_sync_probe.txt): a warmed AlproForPretrain forward + backward queues its launches without a single synchronising
    call -- torch.cuda.set_sync_debug_mode("error") raises on any (the reference's `vtm_labels ... .to(device)`, alpro_models.py:327, was one: a
    pageable upload waits for everything queued on the launch stream).  With the host a whole step ahead, the side streams' launches are queued
    when a CU frees up, and the only syncs of a training loop are the ones the driver asks for."""
    from alpro_amd import amp, config as rt
    m, batch, _ = fixture_models("pretrain_T8")
    was_training = m.training
    sc = amp.LossScaler(init_scale=4096.0, dynamic=False, device="cuda")   # (built before the check: its state is uploaded from a host list)
    try:
        fresh_grads(m)
        for rep in range(3):   # gradients accumulate into the buffers of the first pass, as under FlatAdamW's flat views: job tables keyed on them are built once
            if rep == 2:
                torch.cuda.synchronize()
                torch.cuda.set_sync_debug_mode("error")
            with rt.use_compute_dtype("fp16"), torch.enable_grad():
                rt.set_armed_loss_scaler(sc)
                out = m(batch)
                loss = out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]
                with rt.loss_scaling(sc):
                    (loss * sc.scale.reshape(())).backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
        rt.set_armed_loss_scaler(None)
        torch.cuda.synchronize()
        fresh_grads(m)
        m.train(was_training)
    assert bool(torch.isfinite(loss.detach()).all())


@pytest.mark.parametrize("mode", ["fp16", "fp32"])
def test_text_pass_on_its_side_stream_is_bitwise_neutral(fixture_models, monkeypatch, mode):
    """Round 6 (alpro_amd.config.text_side_stream; AlproForPretrain.forward): the 2B-caption text-encoder pass runs on a side stream beside the visual
    encoder's forward, its backward (autograd: on the stream of the forward) beside the visual encoder's backward; the launch stream waits where the
    text rows are first needed and takes the side stream's parameter gradients over in an end-of-backward callback.  Every output of the pretraining
    forward and every parameter gradient must be bit for bit what the one-stream step gives -- twice each, alternating, on the 8-frame fixture model."""
    from alpro_amd import config as rt
    m, batch, _ = fixture_models("pretrain_T8")
    monkeypatch.setattr(torch, "multinomial", argmax_multinomial)
    monkeypatch.setattr(np.random, "uniform", lambda *a, **k: 1.0)      # use_mask_prob branch not taken, on every pass
    prev, was_training = rt.text_stream_enabled(), m.training
    ref = None
    try:
        for side in (False, True, False, True):
            rt.set_text_stream(side)
            fresh_grads(m)
            with rt.use_compute_dtype(mode), torch.enable_grad():
                sc = arm_scale(mode)
                out = m(batch)
                loss = out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]
                backward(loss, mode)
                del sc
            cur = [out[k].detach().float().clone() for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "mpm_logits")]
            cur.append(torch.cat([p.grad.reshape(-1).float() for p in m.parameters() if p.grad is not None]).clone())
            torch.cuda.synchronize()
            if side:
                assert rt._TEXT_SIDE, "the side stream was never created"
            if ref is None:
                ref = cur
                assert bool(torch.isfinite(ref[-1]).all()) and float(ref[-1].abs().sum()) > 0
            for a, b, what in zip(cur, ref, ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "mpm_logits", "parameter gradients")):
                assert torch.equal(a, b), "%s differs with the text pass on its side stream = %s (max %.3e)" % (what, side, float((a - b).abs().max()))
    finally:
        rt.set_text_stream(prev)
        rt.set_armed_loss_scaler(None)
        fresh_grads(m)
        m.train(was_training)


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_weight_gradients_on_the_side_stream_are_bitwise_neutral(mode):
    """Round 6 (alpro_amd.config.wgrad_side_stream; alpro_amd.modeling.train.wgrad): inside an anchored backward the 16-bit weight-gradient GEMMs
    are launched on a second HIP stream behind an event of the launch stream, which waits for them where the backward returns.  Same kernels, same
    operands, same summation order: every parameter gradient of the encoder's training forward + hand-written backward must be bit for bit what
    the one-stream backward leaves (three backward passes each, the later ones into warm allocator blocks that the side stream has used)."""
    from alpro_amd import config as rt
    from alpro_amd.modeling.timesformer.vit import TimeSformer
    torch.manual_seed(41)
    T, B = 4, 8
    enc = TimeSformer(dict(VENC, num_frm=T, drop_path_rate=0.0), input_format="RGB").cuda().train()
    with torch.no_grad():
        for blk in enc.model.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    x = torch.randn(B, 3, T, 224, 224, device="cuda")
    dout = torch.randn(B, 197, 768, device="cuda") * 1e-2
    prev = rt.wgrad_stream_enabled()
    res = {}
    try:
        for side in (False, True, False, True):
            rt.set_wgrad_stream(side)
            with rt.use_compute_dtype(mode), rt.use_cls_precise("0"):
                for rep in range(3):
                    for p in enc.parameters():
                        p.grad = None
                    sc = arm_scale(mode)
                    with torch.enable_grad():
                        yt = enc.forward_features(x)
                        backward((yt * dout).sum(), mode)
                    del sc
                    g = torch.cat([p.grad.reshape(-1) for p in enc.parameters() if p.grad is not None]).clone()
                    torch.cuda.synchronize()
                    if "ref" not in res:
                        res["ref"] = g
                    assert torch.equal(g, res["ref"]), "parameter gradients differ (side stream %s, pass %d): max %.3e" % (side, rep, float((g - res["ref"]).abs().max()))
            if side:
                ent = [e for e in rt._WGRAD_SIDE.values()]
                assert ent and all(not e[1] for e in ent), "the side stream was not used, or not joined where the backward returned"
    finally:
        rt.set_wgrad_stream(prev)
        rt.set_armed_loss_scaler(None)
    assert bool(torch.isfinite(res["ref"]).all()) and float(res["ref"].abs().sum()) > 0


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_inference_forward_schedules_are_bitwise_neutral(mode):
    """Round 6, two schedule changes of the no-grad encoder forward that must not move a bit:
    (i) alpro_amd.config.defer_temporal_add -- add + norm1 does not write x + temporal branch, add + norm2 adds both branches to the block input
        in the same order of fp32 additions (alpro_add_layernorm_pre_mlp2);
    (ii) alpro_amd.config.split_streams -- the two halves of the batch through every block on two HIP streams (vit.run_blocks), with and without
        the precise CLS chain on its side streams, meeting once behind the last block or at every block boundary.
    forward_features and forward_cls of the same clips, every combination against the one-stream round-3 form.  (B = 16 x 4 frames: the whole batch
    and its halves send every Linear to the same GEMM kernel -- N = 768: 147 / 72 tiles, the 128 x 128 kernel; N = 2304 / 3072: >= 216 tiles, the 8-phase
    kernel -- so every output element is the same sequence of fp32 operations; across the 160-tile switch the two kernels sum in different orders.)"""
    import os
    from alpro_amd import config as rt
    from alpro_amd.modeling.timesformer.vit import TimeSformer
    torch.manual_seed(37)
    T, B = 4, 16
    enc = TimeSformer(dict(VENC, num_frm=T, drop_path_rate=0.0), input_format="RGB").cuda().eval()
    with torch.no_grad():
        for blk in enc.model.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
            torch.nn.init.normal_(blk.temporal_fc.bias, std=0.02)
    x = torch.randn(B, 3, T, 224, 224, device="cuda")
    prev = (rt._defer_tadd[0], rt._split_streams[0], rt._cls_stream[0], os.environ.get("ALPRO_SPLIT_LOCKSTEP"))
    outs = {}
    try:
        for cp in ("0", "1"):
            for defer in (False, True):
                for split, lock, cs in (("0", "0", "0"), ("1", "0", "0"), ("1", "0", "infer"), ("1", "1", "infer")):
                    rt.set_defer_temporal_add(defer)
                    rt.set_split_streams(split)
                    rt.set_cls_stream(cs)
                    os.environ["ALPRO_SPLIT_LOCKSTEP"] = lock
                    with rt.use_compute_dtype(mode), rt.use_cls_precise(cp), torch.no_grad():
                        for rep in range(2):   # twice: the second pass runs with warm operand copies (no fork-after-rebuild event)
                            y = enc.forward_features(x).float().clone()
                            c = enc.forward_cls(x).float().clone()
                    torch.cuda.synchronize()
                    outs[(cp, defer, split, lock, cs)] = (y, c)
    finally:
        rt.set_defer_temporal_add(prev[0])
        rt.set_split_streams(prev[1])
        rt.set_cls_stream(prev[2])
        if prev[3] is None:
            os.environ.pop("ALPRO_SPLIT_LOCKSTEP", None)
        else:
            os.environ["ALPRO_SPLIT_LOCKSTEP"] = prev[3]
    for cp in ("0", "1"):
        ref = outs[(cp, False, "0", "0", "0")]
        assert bool(torch.isfinite(ref[0]).all()) and float(ref[0].abs().sum()) > 0
        for key, (y, c) in outs.items():
            if key[0] != cp:
                continue
            assert torch.equal(y, ref[0]), "forward_features differs for (cls_precise, defer, split, lockstep, cls_stream) = %s" % (key,)
            assert torch.equal(c, ref[1]), "forward_cls differs for %s" % (key,)
