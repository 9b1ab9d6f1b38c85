"""CPU: pin oracle/alpro_oracle.py against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  fp32 vs fp32 on the same closed-form weights/inputs; the only
differences are summation order, so the tolerances are tight (rtol 2e-4 / atol 2e-5)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import alpro_oracle as ao
from tests.golden.det_init import det_batch, det_param, unit_uniform
from tests.conftest import GOLDEN

RT, AT = 2e-4, 2e-5


def close(a, b, rtol=RT, atol=AT, what=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a.astype(np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol, err_msg=what)


def test_state_spec_matches_reference_keys(bert_cfg):
    keys = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    spec = ao.alpro_state_spec("retrieval", bert_cfg, num_frm=2)
    assert {k: list(v) for k, v in spec.items()} == keys["retrieval_T2"]
    spec = ao.alpro_state_spec("pretrain", bert_cfg, num_frm=8)
    assert {k: list(v) for k, v in spec.items()} == keys["pretrain_T8"]


def test_block_droppath_train_mode():
    g = np.load(os.path.join(GOLDEN, "block11_droppath_T2_B4.npz"))
    B, T = 4, 2
    name = "visual_encoder.model.blocks.11"
    p = {k: det_param(k, s) for k, s in ao.alpro_state_spec("retrieval", {"hidden_size": 768, "max_position_embeddings": 8, "vocab_size": 8, "type_vocab_size": 2, "num_hidden_layers": 0, "intermediate_size": 8}, T).items() if k.startswith(name)}
    x = torch.from_numpy(unit_uniform("block_in", B * (1 + 196 * T) * 768).astype(np.float32)).view(B, 1 + 196 * T, 768)
    keep = 0.9  # drop_path 0.1 at layer 11 (vit.py:272); vit_utils.py:146-151
    drop = {k: torch.floor(keep + torch.from_numpy(g["rand_%d" % i])) / keep for i, k in enumerate("tsm")}
    y = ao.vit_block(x, p, name, B, T, 14, drop)
    close(y[:, [0, 1, 2, 200, 392]], g["y_rows"], what="block rows")
    close(y.norm(dim=-1), g["y_rownorm"], what="block row norms")
    assert (drop["t"] == 0).any() and (drop["t"] > 1).any()


@pytest.fixture(scope="module")
def retrieval_case(bert_cfg):
    T, B = 2, 3
    p = ao.det_state("retrieval", bert_cfg, T)
    orc = ao.AlproOracle(p, bert_cfg, T)
    batch = det_batch(B, T, seed_name="retrieval_T2", with_mlm=False, with_mpm=False)
    return orc, batch, np.load(os.path.join(GOLDEN, "retrieval_T2_B3.npz"))


def test_retrieval_forward(retrieval_case):
    orc, batch, g = retrieval_case
    with torch.no_grad():
        out = orc.forward_retrieval(batch)
        ve = orc.visual_embeds(batch["visual_inputs"])
    close(out["itc_loss"], g["itc_loss"]); close(out["itm_loss"], g["itm_loss"])
    close(out["itm_scores"], g["itm_scores"]); close(out["itm_labels"], g["itm_labels"])
    close(ve[:, [0, 1, 100, 196]], g["video_embeds_rows"]); close(ve.norm(dim=-1), g["video_embeds_rownorm"])
    close(ve.sum(dim=-1), g["video_embeds_rowsum"], atol=2e-4)


def test_retrieval_world2_vtc(retrieval_case):
    """VTC targets use local_rank*B offsets into the gathered features (alpro_models.py:121-123)."""
    orc, batch, g = retrieval_case
    B = 3
    ov = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/video", B * 256).astype(np.float32)).view(B, 256), dim=-1)
    ot = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/text", B * 256).astype(np.float32)).view(B, 256), dim=-1)
    with torch.no_grad():
        out = orc.forward_retrieval(batch, world=(1, [ov, None], [ot, None]))
    close(out["itc_loss"], g["w2_itc_loss"]); close(out["itm_loss"], g["w2_itm_loss"]); close(out["itm_scores"], g["w2_itm_scores"])


def test_retrieval_inference(retrieval_case):
    orc, batch, g = retrieval_case
    with torch.no_grad():
        inf = orc.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1], text_input_ids=batch["text_input_ids"],
                                         text_input_mask=batch["text_input_mask"]))
    close(inf["logits"], g["inf_logits"]); close(inf["itc_scores"], g["inf_itc_scores"])


@pytest.mark.slow
def test_pretrain_forward_backward(bert_cfg):
    """All 10 outputs of AlproForPretrain.forward + gradients of loss = mlm+itm+itc+mpm (run_pretrain_sparse.py:557)."""
    T, B = 8, 2
    g = np.load(os.path.join(GOLDEN, "pretrain_T8_B2.npz"))
    skip = ("prompter.text_encoder.", "prompter.itm_head", "prompter.text_proj", "visual_encoder.model.head", "prompter.visual_encoder.model.head")
    spec = ao.alpro_state_spec("pretrain", bert_cfg, T)
    only = [k for k in spec if not k.startswith(skip)]
    p = ao.det_state("pretrain", bert_cfg, T, only=only)
    names = [str(n) for n in g["grad_norm_names"]]
    for n in names:
        p[n].requires_grad_(True)
    orc = ao.AlproOracle(p, bert_cfg, T)
    out = orc.forward_pretrain(det_batch(B, T, seed_name="pretrain_T8"))
    for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "mpm_logits", "mpm_labels", "video_feat", "text_feat",
              "text_embeds", "sim_v2t"):
        close(out[k], g[k], what=k)
    close(out["mlm_scores"][:, :, ::61], g["mlm_scores_cols"], what="mlm cols")
    close(torch.logsumexp(out["mlm_scores"], -1), g["mlm_scores_lse"], what="mlm lse")
    ve = out["video_embeds"]
    close(ve[:, [0, 1, 57, 196]], g["video_embeds_rows"]); close(ve.norm(dim=-1), g["video_embeds_rownorm"])
    loss = out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]
    loss.backward()
    # tied tensors: the reference reports the word-embedding grad once (decoder grad accumulates into it)
    got = np.array([float(p[n].grad.norm()) for n in names])
    np.testing.assert_allclose(got, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("grad/"):
            close(p[k[5:]].grad, g[k], rtol=2e-3, atol=2e-6, what=k)


@pytest.mark.slow
def test_pretrain_released_geometry(bert_cfg):
    """The geometry the reference pretrains with (config_release/pretrain_alpro.json:34,37,59: 4 frames, 30 tokens -> fusion length
    227): all losses, VTC logits, embeddings and the 460 parameter-gradient norms of loss = mlm + itm + itc + mpm."""
    T, B, Lt = 4, 2, 30
    g = np.load(os.path.join(GOLDEN, "pretrain_release_T4_L30_B2.npz"))
    skip = ("prompter.text_encoder.", "prompter.itm_head", "prompter.text_proj", "visual_encoder.model.head", "prompter.visual_encoder.model.head")
    spec = ao.alpro_state_spec("pretrain", bert_cfg, T)
    p = ao.det_state("pretrain", bert_cfg, T, only=[k for k in spec if not k.startswith(skip)])
    names = [str(n) for n in g["grad_norm_names"]]
    for n in names:
        p[n].requires_grad_(True)
    orc = ao.AlproOracle(p, bert_cfg, T)
    out = orc.forward_pretrain(det_batch(B, T, Lt=Lt, seed_name="pretrain_release"))
    assert out["mlm_scores"].shape[:2] == (B, Lt)
    for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "mpm_logits", "sim_v2t"):
        close(out[k], g[k], what=k)
    close(out["mlm_scores"][:, :, ::61], g["mlm_scores_cols"], what="mlm cols")
    close(out["text_embeds"][:, [0, 1, 29]], g["text_embeds_rows"], what="text_embeds rows")
    ve = out["video_embeds"]
    close(ve[:, [0, 1, 57, 196]], g["video_embeds_rows"]); close(ve.norm(dim=-1), g["video_embeds_rownorm"])
    (out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]).backward()
    got = np.array([float(p[n].grad.norm()) for n in names])
    np.testing.assert_allclose(got, g["grad_norms"], rtol=2e-3, atol=1e-7)


def _other_rank_feats(B):
    ov = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/video", B * 256).astype(np.float32)).view(B, 256), dim=-1)
    ot = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/text", B * 256).astype(np.float32)).view(B, 256), dim=-1)
    return ov, ot


def test_prompter_build_prompts_forward_and_pseudo_labels(bert_cfg):
    """Prompter (alpro_models.py:389-632) against the reference run on 8 entities x 12 / 10 templates: build_text_prompts
    buffers, forward (VTC of the teacher, single rank and as rank 1 of 2), get_pseudo_labels on both prompt sets."""
    from tests.golden.det_init import det_prompts
    T, B, E = 2, 3, 8
    g = np.load(os.path.join(GOLDEN, "prompter_T2_B3_E8.npz"))
    p = ao.det_state("prompter", bert_cfg, T, num_entities=E)
    orc = ao.AlproOracle(p, bert_cfg, T)
    pv, pi = det_prompts(E, 12, 15, "prompts/video"), det_prompts(E, 10, 15, "prompts/image")
    vfeat = orc.build_text_prompts(pv.input_ids, pv.attention_mask, E)
    ifeat = orc.build_text_prompts(pi.input_ids, pi.attention_mask, E, step_size=32)   # chunking must not matter
    close(vfeat, g["video_prompt_feat"], what="video_prompt_feat")
    close(ifeat, g["image_prompt_feat"], what="image_prompt_feat")
    assert vfeat.shape == (E, 256) and float(vfeat.norm(dim=-1).max()) < 1.0   # mean of unit vectors
    batch = det_batch(B, T, seed_name="prompter_T2", with_mlm=False, with_mpm=True)
    with torch.no_grad():
        out = orc.forward_prompter(batch)
        ov, ot = _other_rank_feats(B)
        out2 = orc.forward_prompter(batch, world=(1, [ov, None], [ot, None]))
    for k in ("itc_loss", "i2t_scores", "t2i_scores"):
        close(out[k], g[k], what=k)
        close(out2[k], g["w2_" + k], what="w2_" + k)
    assert np.array_equal(out["itc_labels"].numpy(), g["itc_labels"].astype(np.int64))
    assert np.array_equal(out2["itc_labels"].numpy(), g["w2_itc_labels"].astype(np.int64)) and out2["itc_labels"].min() == B
    orc.p["video_prompt_feat"], orc.p["image_prompt_feat"] = vfeat, ifeat
    for ty in ("video", "img"):
        soft, ign = orc.pseudo_labels(batch["crop_visual_inputs"], ty)
        close(soft, g["pseudo_labels_" + ty], what="pseudo labels " + ty)
        assert np.array_equal(ign.numpy().astype(np.float32), g["pseudo_ignore_" + ty])


def test_retrieval_finetune_gradients(bert_cfg):
    """loss = itm_loss + itc_loss (run_video_retrieval.py:432-434) backward through AlproForVideoTextRetrieval.forward:
    per-parameter gradient norms of every trained tensor + 12 full gradients vs the reference's autograd."""
    T, B = 2, 3
    g = np.load(os.path.join(GOLDEN, "retrieval_grads_T2_B3.npz"))
    p = ao.det_state("retrieval", bert_cfg, T)
    names = [str(n) for n in g["grad_norm_names"]]
    for n in names:
        p[n].requires_grad_(True)
    orc = ao.AlproOracle(p, bert_cfg, T)
    out = orc.forward_retrieval(det_batch(B, T, seed_name="retrieval_T2", with_mlm=False, with_mpm=False))
    for k in ("itc_loss", "itm_loss", "itm_scores"):
        close(out[k], g[k], what=k)
    (out["itm_loss"] + out["itc_loss"]).backward()
    got = np.array([0.0 if p[n].grad is None else float(p[n].grad.norm()) for n in names])
    np.testing.assert_allclose(got, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("grad/"):
            close(p[k[5:]].grad, g[k], rtol=2e-3, atol=2e-6, what=k)


def test_retrieval_16_frames(bert_cfg):
    """Model-level forward with a 16-slot time_embed (BASELINE configs[4] quotes 16 frames)."""
    T, B = 16, 2
    g = np.load(os.path.join(GOLDEN, "retrieval_T16_B2.npz"))
    orc = ao.AlproOracle(ao.det_state("retrieval", bert_cfg, T), bert_cfg, T)
    batch = det_batch(B, T, seed_name="retrieval_T16", with_mlm=False, with_mpm=False)
    with torch.no_grad():
        out = orc.forward_retrieval(batch)
        ve = orc.visual_embeds(batch["visual_inputs"])
        inf = orc.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1], text_input_ids=batch["text_input_ids"],
                                         text_input_mask=batch["text_input_mask"]))
    for k in ("itc_loss", "itm_loss", "itm_scores", "itm_labels"):
        close(out[k], g[k], what=k)
    close(ve[:, [0, 1, 100, 196]], g["video_embeds_rows"]); close(ve.norm(dim=-1), g["video_embeds_rownorm"])
    close(inf["logits"], g["inf_logits"]); close(inf["itc_scores"], g["inf_itc_scores"])


def test_optimizer_epilogue_trajectory():
    """a22: the oracle's clip + AdamW + lr schedule against the trajectory the REFERENCE's AdamW / get_lr_sched / torch clip_grad_norm_
    produced (tests/golden/optimizer_adamw_3steps.npz, make_golden.case_optimizer): parameters after each of three steps, the logged
    gradient norms, the learning rates and the final moments, with and without weight decay, clipped and unclipped steps."""
    from tests.golden.det_init import OPT_SCENARIOS, opt_tensors
    g = np.load(os.path.join(GOLDEN, "optimizer_adamw_3steps.npz"))
    for sc, hp in OPT_SCENARIOS.items():
        params = [t.clone() for t in opt_tensors("param")]
        m, v = [torch.zeros_like(t) for t in params], [torch.zeros_like(t) for t in params]
        clipped = []
        for step in range(3):
            lr = ao.lr_sched(step + 1, hp["decay"], hp["lr"], hp["num_train_steps"], hp["warmup_ratio"])
            assert lr == pytest.approx(float(g["%s/lr/%d" % (sc, step)]), rel=1e-12)
            total = ao.clip_and_adamw_step(params, opt_tensors("grad", step), m, v, step + 1, lr, hp["betas"], 1e-6, hp["weight_decay"], hp["grad_norm"])
            assert total == pytest.approx(float(g["%s/grad_norm/%d" % (sc, step)]), rel=1e-6)
            clipped.append(total > hp["grad_norm"])
            close(torch.cat([p.reshape(-1) for p in params]), g["%s/params/%d" % (sc, step)], rtol=1e-6, atol=1e-8, what="%s params after step %d" % (sc, step))
        assert clipped == [True, True, False]
        close(torch.cat([t.reshape(-1) for t in m]), g[sc + "/exp_avg"], rtol=1e-5, atol=1e-9)
        close(torch.cat([t.reshape(-1) for t in v]), g[sc + "/exp_avg_sq"], rtol=1e-5, atol=1e-12)
