"""The four reference-generated fixtures that carry VTC logits, as one table (TEST INFRASTRUCTURE: tests/, bench.py's in-run parity
measurement and tools/ use it; nothing under alpro_amd/ does).

Every case builds the real model class with the closed-form weights of det_init.py, runs the closed-form batch and compares the VTC logits
(video_feat @ text_feat^T / temp, alpro_models.py:103-128 / 893-897) with what the REFERENCE produced for the same weights and inputs
(tests/golden/*.npz, written by make_golden.py from /root/reference).  The north star's bar is on the WORST of them."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
NORTH_STAR_BAR = 1e-3

# name -> (model class name, frames, batch, caption length, det_batch seed name, fixture file, fixture key, how the logits are produced)
CASES = {
    "retrieval_T2": ("AlproForVideoTextRetrieval", 2, 3, 40, "retrieval_T2", "retrieval_T2_B3.npz", "inf_itc_scores", "inference"),
    "pretrain_T8": ("AlproForPretrain", 8, 2, 40, "pretrain_T8", "pretrain_T8_B2.npz", "sim_v2t", "feats"),
    "retrieval_T16": ("AlproForVideoTextRetrieval", 16, 2, 40, "retrieval_T16", "retrieval_T16_B2.npz", "inf_itc_scores", "inference"),
    "pretrain_release_T4_L30": ("AlproForPretrain", 4, 2, 30, "pretrain_release", "pretrain_release_T4_L30_B2.npz", "sim_v2t", "feats"),
}


def build_case(name, bert_cfg, venc, make_cfg, device):
    """-> (model in eval mode on `device`, batch on `device`, reference logits as float64 numpy)."""
    from alpro_amd.modeling import alpro_models as am
    from tests.golden.det_init import det_batch, fill_state_dict_
    cls, T, B, Lt, seed, fname, key, how = CASES[name]
    m = getattr(am, cls)(make_cfg(bert_cfg), dict(venc, num_frm=T))
    fill_state_dict_(m)
    m.eval().to(device)
    full = cls == "AlproForPretrain"
    batch = det_batch(B, T, Lt=Lt, seed_name=seed, with_mlm=full, with_mpm=full)
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    ref = np.asarray(np.load(os.path.join(HERE, fname))[key], dtype=np.float64)
    return m, batch, ref


def vtc_logits(name, m, batch):
    how = CASES[name][7]
    with torch.no_grad():
        if how == "inference":   # 1 video x n captions (alpro_models.py:874-914)
            return m.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1], text_input_ids=batch["text_input_ids"],
                                            text_input_mask=batch["text_input_mask"]))["itc_scores"]
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        _, tf = m._forward_text_feats(batch)
        return m._video_feat(ve) @ tf.t() / m.temp


def vtc_logit_error(name, m, batch, ref):
    got = vtc_logits(name, m, batch).detach().float().cpu().numpy().astype(np.float64)
    return float(np.abs(got - ref).max())


# what tests/test_model_parity.py asserts and bench.py states next to its measurement (fp16 operands + precise CLS rows against the exact mode,
# B = 64 x 8 frames): about twice the first measurement (profiles/r5_parity_backward_B64.txt: l2 rel err worst 4.7e-3 / median 3.0e-3, cosine worst
# 0.999989, norm rel err worst 9.0e-4, global cosine 0.999997, losses <= 2.8e-4)
FULL_SIZE_BACKWARD_LIMITS = dict(grad_l2_rel_err_worst=1e-2, grad_l2_rel_err_median=6e-3, grad_norm_rel_err_worst=3e-3, grad_norm_rel_err_median=5e-4,
                                 grad_cosine_worst=0.9999, grad_cosine_median=0.99998, global_grad_cosine=0.99999, global_grad_norm_rel_err=5e-4, loss_abs_err=1e-3)


def full_size_backward_parity(bert_cfg, venc, make_cfg, device, dtype="fp16", cls_precise="auto", B=64, T=8, seed=4, loss_scale=65536.0, model="pretrain"):
    """Parity of the BACKWARD at the benchmarked size.  No golden vectors exist there; the exact fp32 HIP mode -- pinned to
    the reference's gradients at <= 2e-3 on the fixtures (tests/test_model_parity.py) -- is the oracle.  One training step's forward + backward
    of AlproForPretrain (VTC + VTM + MLM + MPM; B pairs x T frames x 224^2 + 40 tokens; train mode with drop-path and dropout at 0 so that both
    modes differentiate the same function; the hard negatives of the exact run are re-used) in the exact mode and in `dtype` (fp16: a loss-scaled
    backward at 2^16, like the timed steps; run_pretrain_sparse.py:557,595-601), then per parameter tensor:
        norm_rel = | |g| - |g_exact| | / |g_exact|,   cos = <g, g_exact> / (|g| |g_exact|),   l2_rel = |g - g_exact| / |g_exact|
    -> dict with worst / median of each over all tensors that receive a gradient, the worst tensors' names and the four losses' errors.
    model="retrieval" (round 6): the FINETUNE step of BASELINE configs[4] instead -- AlproForVideoTextRetrieval, loss = itm_loss +
    itc_loss (run_video_retrieval.py:432-434), at config_release/msrvtt_ret.json's geometry when called with B = 8, T = 8 (num_frm 8,
    train_batch_size 8, max_txt_len 40)."""
    import bench
    from alpro_amd import amp, config as rt
    from alpro_amd.modeling.alpro_models import AlproBaseModel, AlproForPretrain, AlproForVideoTextRetrieval
    torch.manual_seed(seed)
    cfg = make_cfg(dict(bert_cfg, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    Model = AlproForPretrain if model == "pretrain" else AlproForVideoTextRetrieval
    m = Model(cfg, dict(venc, num_frm=T, drop_path_rate=0.0)).to(device).train()
    with torch.no_grad():   # TimeSformer zero-initialises temporal_fc of blocks 1..11 (vit.py:306-315): the temporal halves would get exactly-zero gradients
        for blk in m.visual_encoder.model.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    batch = bench.synth_batch(B, T, device, seed=11, full=model == "pretrain")
    batch["text_input_mask"] = batch["text_input_mask"].clone()
    batch["text_input_mask"][::3, 31:] = 0
    negs = {}
    orig_neg, orig_mn = AlproBaseModel.__dict__["_sample_negatives"], torch.multinomial   # (the staticmethod object of the shared base class)

    def record(sim_v2t, sim_t2v, bs):
        if "n" not in negs:
            negs["n"] = orig_neg.__func__(sim_v2t, sim_t2v, bs)
        return negs["n"]
    AlproBaseModel._sample_negatives = staticmethod(record)
    torch.multinomial = lambda w, n=1, *a, **k: w.argmax(dim=-1, keepdim=True)
    prev_armed = rt._armed[0]
    keys = ("mlm_loss", "itm_loss", "itc_loss", "mpm_loss") if model == "pretrain" else ("itm_loss", "itc_loss")

    used_cls = {}

    def run(dt, cls):
        for p in m.parameters():
            p.grad = None
        with rt.use_compute_dtype(dt), rt.use_cls_precise(cls), torch.enable_grad():
            used_cls[dt] = bool(rt.cls_precise())
            sc, scale = None, 1.0
            if amp.needs_loss_scaling():   # armed BEFORE the forward, like the optimizer's scaler in the timed steps (the LM head writes its logit gradient at forward time)
                sc, scale = amp.LossScaler(init_scale=loss_scale, dynamic=False, device=device), loss_scale
            rt.set_armed_loss_scaler(sc)
            out = m(batch)
            loss = sum(out[k] for k in keys)
            if sc is not None:
                with rt.loss_scaling(sc):
                    (loss * sc.scale.reshape(())).backward()
            else:
                loss.backward()
        grads = {n: (p.grad.detach().double() / scale) for n, p in m.named_parameters() if p.grad is not None and (p.grad.dim() == 0 or any(p.grad.stride()) or p.grad.numel() <= 1)}
        return {k: float(out[k]) for k in keys}, grads
    try:
        l32, g32 = run("fp32", "auto")
        l16, g16 = run(dtype, cls_precise)
    finally:
        AlproBaseModel._sample_negatives = orig_neg
        torch.multinomial = orig_mn
        rt._armed[0] = prev_armed
    assert set(g16) == set(g32), sorted(set(g16) ^ set(g32))[:5]
    rows, nonfinite = [], []
    for n in g32:
        a, b = g16[n].reshape(-1), g32[n].reshape(-1)
        if not (bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())):
            nonfinite.append((n, int((~torch.isfinite(a)).sum()), int((~torch.isfinite(b)).sum())))
            continue
        nb = float(b.norm())
        if nb == 0.0 or n.endswith("attention.self.key.bias"):   # exactly 0 in exact arithmetic (softmax shift invariance): rounding noise only
            continue
        na = float(a.norm())
        rows.append((n, abs(na - nb) / nb, float(a @ b) / max(na * nb, 1e-300), float((a - b).norm()) / nb, nb))
    rows.sort(key=lambda r: -r[3])
    import statistics as st
    rep = dict(batch=B, frames=T, mode="%s operands%s" % (dtype, " + precise CLS rows (fp32)" if used_cls.get(dtype) else ""), oracle="the exact fp32 HIP mode on the same weights, inputs and hard negatives",
               grad_tensors=len(rows), nonfinite_tensors=nonfinite, loss_scale=loss_scale if dtype == "fp16" else 1.0,
               grad_norm_rel_err_worst=max(r[1] for r in rows), grad_norm_rel_err_median=st.median(r[1] for r in rows),
               grad_cosine_worst=min(r[2] for r in rows), grad_cosine_median=st.median(r[2] for r in rows),
               grad_l2_rel_err_worst=rows[0][3], grad_l2_rel_err_median=st.median(r[3] for r in rows),
               worst_tensors=[(r[0], float("%.3e" % r[3])) for r in rows[:3]],
               loss_abs_err={k: abs(l16[k] - l32[k]) for k in keys}, losses_exact=l32)
    gt = torch.cat([g32[r[0]].reshape(-1) for r in rows]), torch.cat([g16[r[0]].reshape(-1) for r in rows])
    rep["global_grad_norm_rel_err"] = abs(float(gt[1].norm()) - float(gt[0].norm())) / float(gt[0].norm())
    rep["global_grad_cosine"] = float(gt[0] @ gt[1]) / (float(gt[0].norm()) * float(gt[1].norm()))
    del m, g32, g16, gt
    torch.cuda.empty_cache()
    return rep


def full_size_forward_parity(bert_cfg, venc, make_cfg, device, dtype="fp16", cls_precise="auto", B=64, T=8, seed=4):
    """VTC logits at the benchmarked size: AlproForPretrain, eval mode, B pairs x T frames x 224^2 + 40 tokens, random-init
    weights; ALL B x B video-text logits (alpro_models.py:113-118: video_feat @ text_feat.T / temp) in `dtype` against the exact fp32 HIP mode --
    the oracle at this size, pinned to the reference at <= 5e-6 on every fixture.  -> max / p99.9 / rms of the absolute error and whether the
    north star's 1e-3 holds for the MAXIMUM."""
    import bench
    from alpro_amd import config as rt
    from alpro_amd.modeling.alpro_models import AlproForPretrain
    torch.manual_seed(seed)
    m = AlproForPretrain(make_cfg(bert_cfg), dict(venc, num_frm=T)).eval().to(device)
    batch = bench.synth_batch(B, T, device, seed=11, full=True)
    batch["text_input_mask"] = batch["text_input_mask"].clone()
    batch["text_input_mask"][::3, 31:] = 0

    def logits(dt, cls):
        with rt.use_compute_dtype(dt), rt.use_cls_precise(cls), torch.no_grad():
            ve = m._forward_visual_embeds(batch["visual_inputs"])
            _, tf = m._forward_text_feats(batch)
            return (m._video_feat(ve) @ tf.t() / m.temp).double().cpu()
    ref = logits("fp32", "auto")
    e = (logits(dtype, cls_precise) - ref).abs().flatten()
    rep = dict(batch=B, frames=T, logits=int(e.numel()), oracle="the exact fp32 HIP mode on the same weights and inputs",
               vtc_logits_max_abs_err=float("%.3e" % float(e.max())), vtc_logits_p999_abs_err=float("%.3e" % float(e.kthvalue(max(1, int(0.999 * e.numel())))[0])),
               vtc_logits_rms_err=float("%.3e" % float(e.pow(2).mean().sqrt())), north_star_bar=NORTH_STAR_BAR)
    rep["max_meets_bar"] = bool(rep["vtc_logits_max_abs_err"] <= NORTH_STAR_BAR)
    del m
    torch.cuda.empty_cache()
    return rep


def sampler_property_check(device, world, rank, draws, monkeypatch):
    """The un-patched hard-negative sampler (alpro_amd/modeling/alpro_models.py::_sample_negatives) against the reference's semantics
    (alpro_models.py:287-313): own-rank block, never the positive, frequencies ~ softmax of the similarities.  Used by the GPU test
    (tests/test_model_parity.py) and, on the CPU, by tests/test_host_cpu.py."""
    from alpro_amd import dist
    from alpro_amd.modeling.alpro_models import AlproBaseModel
    bs = 16
    g = torch.Generator(device="cpu").manual_seed(5 + world)
    sim_v2t = (torch.randn(bs, bs * world, generator=g) * 2.0).to(device)
    sim_t2v = (torch.randn(bs, bs * world, generator=g) * 2.0).to(device)
    monkeypatch.setattr(dist, "local_rank", lambda: rank)
    torch.manual_seed(1234)
    v0, t0 = sim_v2t.clone(), sim_t2v.clone()
    cnt_v = torch.zeros(bs, bs, dtype=torch.long, device=device)
    cnt_t = torch.zeros(bs, bs, dtype=torch.long, device=device)
    rows = torch.arange(bs, device=device)
    for _ in range(draws):
        neg_video, neg_text = AlproBaseModel._sample_negatives(sim_v2t, sim_t2v, bs)
        assert neg_video.shape == (bs,) and neg_text.shape == (bs,) and neg_video.dtype == torch.long
        cnt_v[rows, neg_video] += 1       # a negative video for each text: drawn from sim_t2v's own-rank block
        cnt_t[rows, neg_text] += 1
    assert torch.equal(sim_v2t, v0) and torch.equal(sim_t2v, t0), "the sampler must not write into the similarity matrices (the loss still needs them)"
    for cnt, sim, what in ((cnt_v, sim_t2v, "negative video per text"), (cnt_t, sim_v2t, "negative text per video")):
        assert int(cnt.diagonal().sum()) == 0, "%s: the positive was drawn" % what
        assert int(cnt.sum()) == bs * draws and int(cnt.min()) >= 0
        w = sim[:, bs * rank:bs * (rank + 1)].double().clone()
        w.fill_diagonal_(-float("inf"))
        pr = torch.softmax(w, dim=1)
        freq = cnt.double() / draws
        sigma = (pr * (1 - pr) / draws).sqrt()
        z = ((freq - pr).abs() / sigma.clamp_min(1e-9))
        z[pr == 0] = 0.0
        assert float(z.max()) < 4.0, "%s: a frequency is %.1f sigma from its softmax weight" % (what, float(z.max()))
        # the other rank's block is never a candidate: with world 2 the weights of the wrong block would give a different table
        if world == 2:
            other = sim[:, bs * (1 - rank):bs * (2 - rank)].double().clone()
            other.fill_diagonal_(-float("inf"))
            assert float(((freq - torch.softmax(other, dim=1)).abs() / sigma.clamp_min(1e-9)).max()) > 8.0
