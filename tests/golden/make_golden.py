"""Generate golden vectors by running the REFERENCE itself (CPU, fp32, eval) in this container.

    python -m tests.golden.make_golden            # writes tests/golden/*.npz + state_keys.json

The reference has no tests/fixtures of its own, so these files are the
pin for oracle/alpro_oracle.py.  Weights and inputs are NOT stored: both sides regenerate
them from tests/golden/det_init.py closed forms.  Stochastic ops are pinned:
  * torch.multinomial (hard negatives, alpro_models.py:303,311) -> argmax of the weights,
  * torch.rand inside drop_path (vit_utils.py:148) -> det_init.unit_uniform stream, recorded.
Needs /root/reference; never runs on the GPU box.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.det_init import (OPT_SCENARIOS, det_batch, det_caption_ids, det_prompts, det_raw_clips, fill_state_dict_, opt_tensors,  # noqa: E402
                                    unit_uniform)
from tests.golden import ref_harness as rh  # noqa: E402

MLM_COL_STRIDE = 61


def npf(t):
    return t.detach().to(torch.float32).cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def summarize_embeds(name, t, rows, out):
    out[name + "_rows"] = npf(t[:, rows])
    out[name + "_rownorm"] = npf(t.norm(dim=-1))
    out[name + "_rowsum"] = npf(t.sum(dim=-1))


def argmax_multinomial(w, n=1, *a, **k):
    assert n == 1
    return w.argmax(dim=-1, keepdim=True)


GRAD_FULL = ["vision_proj.weight", "text_proj.bias", "itm_head.weight", "temp",
             "visual_encoder.model.blocks.0.temporal_fc.bias", "visual_encoder.model.blocks.11.attn.qkv.bias",
             "visual_encoder.model.blocks.5.temporal_attn.qkv.bias", "visual_encoder.model.time_embed",
             "visual_encoder.model.cls_token", "visual_encoder.model.blocks.3.norm1.weight",
             "text_encoder.bert.encoder.layer.0.attention.self.query.bias",
             "text_encoder.bert.encoder.layer.11.output.LayerNorm.weight",
             "text_encoder.bert.embeddings.LayerNorm.bias", "text_encoder.cls.predictions.transform.dense.bias",
             "mpm_head.2.bias"]


def case_pretrain(am, T, B, fname, with_grads):
    cfg, venc = rh.make_configs(num_frm=T)
    torch.manual_seed(0)
    m = am.AlproForPretrain(cfg, venc)
    fill_state_dict_(m)
    m.eval()
    batch = det_batch(B, T, seed_name="pretrain_T%d" % T)
    orig = torch.multinomial
    torch.multinomial = argmax_multinomial
    try:
        out = m(batch)
    finally:
        torch.multinomial = orig
    g = {}
    for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "itm_labels", "mpm_logits", "mpm_labels"):
        g[k] = npf(out[k])
    g["mlm_scores_cols"] = npf(out["mlm_scores"][:, :, ::MLM_COL_STRIDE])
    g["mlm_scores_lse"] = npf(torch.logsumexp(out["mlm_scores"], dim=-1))
    g["mlm_scores_max"] = npf(out["mlm_scores"].max(dim=-1)[0])
    with torch.no_grad():
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        te, tf = m._forward_text_feats(batch)
        summarize_embeds("video_embeds", ve, [0, 1, 57, 196], g)
        g["text_embeds"] = npf(te)
        g["text_feat"] = npf(tf)
        vf = torch.nn.functional.normalize(m.vision_proj(ve[:, 0, :]), dim=-1)
        g["video_feat"] = npf(vf)
        g["sim_v2t"] = npf(vf @ tf.t() / m.temp)
        soft, ign = m.get_pseudo_labels(batch)
        g["pseudo_ignore"] = npf(ign.to(torch.float32))
    if with_grads:
        loss = out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]  # run_pretrain_sparse.py:557
        loss.backward()
        names, norms = [], []
        for n_, p_ in m.named_parameters():
            if p_.grad is not None:
                names.append(n_)
                norms.append(float(p_.grad.norm()))
        g["grad_norm_names"] = np.array(names)
        g["grad_norms"] = np.array(norms, dtype=np.float64)
        pd = dict(m.named_parameters())
        for n_ in GRAD_FULL:
            assert pd[n_].grad is not None, n_
            g["grad/" + n_] = npf(pd[n_].grad)
    np.savez_compressed(os.path.join(HERE, fname), **g)
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    return keys


def case_pretrain_release(am, fname, T=4, Lt=30, B=2):
    """The geometry the reference actually pretrains with (config_release/pretrain_alpro.json:34,37,59: num_frm 4, max_txt_len 30,
    train_batch_size 16): AlproForPretrain at 4 frames x 30 tokens (fusion length 227), all losses + parameter-gradient norms."""
    cfg, venc = rh.make_configs(num_frm=T)
    m = am.AlproForPretrain(cfg, venc)
    fill_state_dict_(m)
    m.eval()
    batch = det_batch(B, T, Lt=Lt, seed_name="pretrain_release")
    orig = torch.multinomial
    torch.multinomial = argmax_multinomial
    try:
        out = m(batch)
    finally:
        torch.multinomial = orig
    g = {k: npf(out[k]) for k in ("itc_loss", "itm_loss", "mlm_loss", "mpm_loss", "itm_scores", "itm_labels", "mpm_logits")}
    g["mlm_scores_cols"] = npf(out["mlm_scores"][:, :, ::MLM_COL_STRIDE])
    with torch.no_grad():
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        te, tf = m._forward_text_feats(batch)
        summarize_embeds("video_embeds", ve, [0, 1, 57, 196], g)
        vf = torch.nn.functional.normalize(m.vision_proj(ve[:, 0, :]), dim=-1)
        g["sim_v2t"] = npf(vf @ tf.t() / m.temp)
        g["text_embeds_rows"] = npf(te[:, [0, 1, 29]])
    (out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]).backward()
    names, norms = [], []
    for n_, p_ in m.named_parameters():
        if p_.grad is not None:
            names.append(n_)
            norms.append(float(p_.grad.norm()))
    g["grad_norm_names"] = np.array(names)
    g["grad_norms"] = np.array(norms, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, fname), **g)


def case_retrieval(am, T, B, fname):
    cfg, venc = rh.make_configs(num_frm=T)
    m = am.AlproForVideoTextRetrieval(cfg, venc)
    fill_state_dict_(m)
    m.eval()
    batch = det_batch(B, T, seed_name="retrieval_T%d" % T, with_mlm=False, with_mpm=False)
    g = {}
    orig = torch.multinomial
    torch.multinomial = argmax_multinomial
    try:
        with torch.no_grad():
            out = m(batch)
            for k in ("itc_loss", "itm_loss", "itm_scores", "itm_labels"):
                g[k] = npf(out[k])
            # world-size-2 VTC semantics: this process plays rank 1; rank 0's features are det tensors
            ve = m.visual_encoder.forward_features(batch["visual_inputs"].transpose(1, 2))
            other_v = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/video", B * 256).astype(np.float32)).view(B, 256), dim=-1)
            other_t = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/text", B * 256).astype(np.float32)).view(B, 256), dim=-1)
            rh.set_sim_ranks(1, [other_v, None], [other_t, None])
            out2 = m(batch)
            rh.clear_sim_ranks()
            for k in ("itc_loss", "itm_loss", "itm_scores"):
                g["w2_" + k] = npf(out2[k])
            summarize_embeds("video_embeds", ve, [0, 1, 100, 196], g)
            # 1 video x B captions inference (alpro_models.py:874-914)
            inf = m.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1],
                                           text_input_ids=batch["text_input_ids"],
                                           text_input_mask=batch["text_input_mask"]))
            g["inf_logits"] = npf(inf["logits"])
            g["inf_itc_scores"] = npf(inf["itc_scores"])
    finally:
        torch.multinomial = orig
        rh.clear_sim_ranks()
    np.savez_compressed(os.path.join(HERE, fname), **g)
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def case_block_droppath(fname, T=2, B=4, layer=11):
    """One ViT Block in TRAIN mode with drop_path=0.1 and a recorded torch.rand stream (vit.py:136-213)."""
    import src.modeling.timesformer.vit as vit
    from functools import partial
    from tests.golden.det_init import det_param
    blk = vit.Block(dim=768, num_heads=12, layer_num=layer, mlp_ratio=4, qkv_bias=True, drop=0., attn_drop=0.,
                    drop_path=0.1, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    with torch.no_grad():
        for k, t in blk.state_dict().items():
            t.copy_(det_param("visual_encoder.model.blocks.%d.%s" % (layer, k), t.shape))
    blk.train()
    x = torch.from_numpy(unit_uniform("block_in", B * (1 + 196 * T) * 768).astype(np.float32)).view(B, 1 + 196 * T, 768)
    calls = []
    orig = torch.rand

    def det_rand(shape, dtype=None, device=None):
        n = int(np.prod(shape))
        r = torch.from_numpy(((unit_uniform("droppath/%d" % len(calls), n) + 1.0) * 0.5).astype(np.float32)).view(*shape)
        calls.append(r.flatten().numpy().copy())
        return r

    torch.rand = det_rand
    try:
        with torch.no_grad():
            y = blk(x, B, T, 14)
    finally:
        torch.rand = orig
    g = {"rand_%d" % i: c for i, c in enumerate(calls)}
    g["y_rows"] = npf(y[:, [0, 1, 2, 200, 392]])
    g["y_rownorm"] = npf(y.norm(dim=-1))
    np.savez_compressed(os.path.join(HERE, fname), **g)


def case_prompter(am, T, B, E, fname, n_templates=12, Lp=15):
    """Prompter (alpro_models.py:389-632): build_text_prompts on E entities x n_templates templates, forward (VTC of the
    teacher, single rank and as rank 1 of a simulated 2-rank job), get_pseudo_labels with the built prompts."""
    cfg, venc = rh.make_configs(num_frm=T, num_entities=E)
    m = am.Prompter(cfg, venc)
    fill_state_dict_(m)
    m.eval()
    prompts = dict(batch_enc_video_prompts=det_prompts(E, n_templates, Lp, "prompts/video"),
                   batch_enc_image_prompts=det_prompts(E, n_templates - 2, Lp, "prompts/image"))
    g = {}
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # build_text_prompts hard-codes .cuda() (alpro_models.py:453)
    try:
        m.build_text_prompts(prompts)
    finally:
        torch.Tensor.cuda = cuda
    g["video_prompt_feat"], g["image_prompt_feat"] = npf(m.video_prompt_feat), npf(m.image_prompt_feat)
    batch = det_batch(B, T, seed_name="prompter_T%d" % T, with_mlm=False, with_mpm=True)
    with torch.no_grad():
        out = m(batch)
        for k in ("itc_loss", "itc_labels", "i2t_scores", "t2i_scores"):
            g[k] = npf(out[k])
        other_v = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/video", B * 256).astype(np.float32)).view(B, 256), dim=-1)
        other_t = torch.nn.functional.normalize(torch.from_numpy(unit_uniform("w2/text", B * 256).astype(np.float32)).view(B, 256), dim=-1)
        rh.set_sim_ranks(1, [other_v, None], [other_t, None])
        try:
            out2 = m(batch)
        finally:
            rh.clear_sim_ranks()
        for k in ("itc_loss", "itc_labels", "i2t_scores", "t2i_scores"):
            g["w2_" + k] = npf(out2[k])
        for ty in ("video", "img"):
            soft, ign = m.get_pseudo_labels(dict(batch, type=ty))
            g["pseudo_labels_" + ty] = npf(soft)
            g["pseudo_ignore_" + ty] = npf(ign.to(torch.float32))
    np.savez_compressed(os.path.join(HERE, fname), **g)


RET_GRAD_FULL = ["vision_proj.weight", "text_proj.bias", "itm_head.weight", "itm_head.bias", "temp",
                 "visual_encoder.model.blocks.11.attn.proj.bias", "visual_encoder.model.blocks.0.temporal_attn.proj.bias",
                 "visual_encoder.model.time_embed", "visual_encoder.model.blocks.6.norm2.weight",
                 "text_encoder.bert.encoder.layer.5.attention.self.value.bias",
                 "text_encoder.bert.encoder.layer.6.attention.output.LayerNorm.weight",
                 "text_encoder.bert.embeddings.LayerNorm.weight"]


def case_retrieval_grads(am, T, B, fname):
    """Retrieval finetune step (BASELINE configs[4]): loss = itm_loss + itc_loss (run_video_retrieval.py:432-434) backward
    through AlproForVideoTextRetrieval.forward (alpro_models.py:733-798); eval-mode layers so the graph is deterministic."""
    cfg, venc = rh.make_configs(num_frm=T)
    m = am.AlproForVideoTextRetrieval(cfg, venc)
    fill_state_dict_(m)
    m.eval()
    batch = det_batch(B, T, seed_name="retrieval_T%d" % T, with_mlm=False, with_mpm=False)
    orig = torch.multinomial
    torch.multinomial = argmax_multinomial
    try:
        out = m(batch)
    finally:
        torch.multinomial = orig
    g = {k: npf(out[k]) for k in ("itc_loss", "itm_loss", "itm_scores")}
    (out["itm_loss"] + out["itc_loss"]).backward()
    names, norms = [], []
    for n_, p_ in m.named_parameters():
        if p_.grad is not None:
            names.append(n_)
            norms.append(float(p_.grad.norm()))
    g["grad_norm_names"] = np.array(names)
    g["grad_norms"] = np.array(norms, dtype=np.float64)
    pd = dict(m.named_parameters())
    for n_ in RET_GRAD_FULL:
        assert pd[n_].grad is not None, n_
        g["grad/" + n_] = npf(pd[n_].grad)
    np.savez_compressed(os.path.join(HERE, fname), **g)


def case_retrieval_frames(am, T, B, fname):
    """Model-level forward with a T-slot time_embed (BASELINE configs[4] quotes 16 frames; config_release/msrvtt_ret.json
    itself sets num_frm 8): forward, visual embeddings and 1-video-x-n-captions inference."""
    cfg, venc = rh.make_configs(num_frm=T)
    m = am.AlproForVideoTextRetrieval(cfg, venc)
    fill_state_dict_(m)
    m.eval()
    batch = det_batch(B, T, seed_name="retrieval_T%d" % T, with_mlm=False, with_mpm=False)
    g = {}
    orig = torch.multinomial
    torch.multinomial = argmax_multinomial
    try:
        with torch.no_grad():
            out = m(batch)
            ve = m.visual_encoder.forward_features(batch["visual_inputs"].transpose(1, 2))
            inf = m.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1], text_input_ids=batch["text_input_ids"],
                                           text_input_mask=batch["text_input_mask"]))
    finally:
        torch.multinomial = orig
    for k in ("itc_loss", "itm_loss", "itm_scores", "itm_labels"):
        g[k] = npf(out[k])
    summarize_embeds("video_embeds", ve, [0, 1, 100, 196], g)
    g["inf_logits"], g["inf_itc_scores"] = npf(inf["logits"]), npf(inf["itc_scores"])
    np.savez_compressed(os.path.join(HERE, fname), **g)


def reference_eval_functions():
    """get_retrieval_metric_from_bool_matrix / get_retrieval_scores / eval_retrieval of the reference driver, EXECUTED from its
    source without importing the module (src/tasks/run_video_retrieval.py pulls in lmdb / decord / cv2 datasets that this image
    lacks): the three function definitions are cut out with `ast` and compiled as they stand."""
    import ast
    from collections import defaultdict
    path = os.path.join(rh.REF, "src/tasks/run_video_retrieval.py")
    tree = ast.parse(open(path).read())
    want = {"get_retrieval_metric_from_bool_matrix", "get_retrieval_scores", "eval_retrieval"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {"np": np, "torch": torch, "defaultdict": defaultdict}
    exec(compile(mod, path, "exec"), ns)
    return ns


def case_retrieval_eval(am, T, V, fname, eval_bsz=3):
    """The reference's retrieval evaluation on V videos x V captions (caption i belongs to video i): records built exactly like
    inference_retrieval (run_video_retrieval.py:642-690: one video, caption mini-batches of eval_bsz, forward_inference, softmax
    of the ITM logits, rounding to 4 decimals) and the metrics its own eval_retrieval computes from them (:558-628)."""
    import torch.nn.functional as F
    cfg, venc = rh.make_configs(num_frm=T)
    m = am.AlproForVideoTextRetrieval(cfg, venc)
    fill_state_dict_(m)
    m.eval()
    batch = det_batch(V, T, seed_name="retrieval_eval_T%d" % T, with_mlm=False, with_mpm=False)
    recs = []
    with torch.no_grad():
        for v in range(V):
            for i in range(0, V, eval_bsz):
                out = m.forward_inference(dict(visual_inputs=batch["visual_inputs"][v:v + 1], text_input_ids=batch["text_input_ids"][i:i + eval_bsz],
                                               text_input_mask=batch["text_input_mask"][i:i + eval_bsz]))
                logits = torch.stack([out["logits"].cpu()]).squeeze().float()
                sims = torch.stack([out["itc_scores"].cpu()]).squeeze().float().tolist()
                if not isinstance(sims, list):
                    sims = [sims]
                if logits.dim() == 1:
                    logits = logits[None]
                probs = F.softmax(logits, dim=1)[:, 1].tolist()
                for j, (p, s) in enumerate(zip(probs, sims)):
                    recs.append(dict(vid_id="v%d" % v, txt_id="t%d" % (i + j), score=round(p, 4), sim=round(s, 4)))
    ns = reference_eval_functions()
    gt = {"t%d" % i: "v%d" % i for i in range(V)}
    metrics = ns["eval_retrieval"](recs, gt, None)
    g = {"score": np.array([r["score"] for r in recs]).reshape(V, V), "sim": np.array([r["sim"] for r in recs]).reshape(V, V)}
    for d in ("text2video", "video2text"):
        for k, val in metrics[d].items():
            g["%s/%s" % (d, k)] = np.float64(val)
    # known-answer check of the metric code itself on a synthetic score table with ties (rounded scores tie often)
    rng = np.random.RandomState(0)
    n = 12
    table = np.round(rng.rand(n, n), 1)
    srecs = [dict(vid_id="v%d" % v, txt_id="t%d" % t, score=float(table[v, t]), sim=0.0) for v in range(n) for t in range(n)]
    sm = ns["eval_retrieval"](srecs, {"t%d" % i: "v%d" % i for i in range(n)}, None)
    g["synthetic_table"] = table
    for d in ("text2video", "video2text"):
        for k, val in sm[d].items():
            g["synthetic/%s/%s" % (d, k)] = np.float64(val)
    np.savez_compressed(os.path.join(HERE, fname), **g)


def reference_input_functions():
    """random_erase (src/datasets/dataset_pretrain_sparse.py:277-311), ImageNorm and mask_batch_text_tokens (src/datasets/data_utils.py:
    437-457, 23-70) of the reference, EXECUTED from their source without importing the modules (which pull in lmdb / decord / cv2 /
    torchvision, absent from this image): the definitions are cut out with `ast` and compiled as they stand."""
    import ast
    ns = {"np": np, "torch": torch}
    for rel, want in (("src/datasets/dataset_pretrain_sparse.py", {"random_erase"}), ("src/datasets/data_utils.py", {"ImageNorm", "mask_batch_text_tokens"})):
        path = os.path.join(rh.REF, rel)
        tree = ast.parse(open(path).read())
        mod = ast.Module(body=[n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want], type_ignores=[])
        assert {n.name for n in mod.body} == want, (rel, want)
        exec(compile(mod, path, "exec"), ns)
    return ns


class _BertUncasedTokenizerStub:
    """What mask_batch_text_tokens asks of the tokenizer (data_utils.py:31-66), with bert-base-uncased's special ids ([PAD] 0, [UNK] 100,
    [CLS] 101, [SEP] 102, [MASK] 103; transformers' get_special_tokens_mask(already_has_special_tokens=True) flags all_special_ids)."""
    mask_token, _pad_token, pad_token_id = "[MASK]", "[PAD]", 0

    def get_special_tokens_mask(self, val, already_has_special_tokens=True):
        return [1 if t in (0, 100, 101, 102, 103) else 0 for t in val]

    def convert_tokens_to_ids(self, tok):
        assert tok == "[MASK]"
        return 103

    def __len__(self):
        return 30522


INPUT_NP_SEED, INPUT_TORCH_SEED, INPUT_STRIDE = 777, 4321, 7


def case_input_pipeline(fname, B=4, T=2):
    """SURVEY 8(f) N4 pinned to the reference: PretrainCollator's per-sample random_erase on the raw uint8 clips
    (dataset_pretrain_sparse.py:244, under np.random.seed), PrefetchLoader's .float() + ImageNorm on visual / crop / context
    (dataloader.py:104-115), and mask_batch_text_tokens under torch.manual_seed -- outputs of the reference's own code."""
    ns = reference_input_functions()
    raw = det_raw_clips(B, T)
    np.random.seed(INPUT_NP_SEED)
    elems = [ns["random_erase"](e, patch_size=16) for e in raw.clone()]
    crop, masks, ctx = (torch.stack([e[i] for e in elems]) for i in range(3))
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self   # ImageNorm.__init__ hard-codes .cuda() (data_utils.py:441-442)
    try:
        norm = ns["ImageNorm"](mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])   # src/configs/config.py defaults
    finally:
        torch.Tensor.cuda = cuda
    g = {"mpm_mask": npf(masks), "np_seed": np.int64(INPUT_NP_SEED), "torch_seed": np.int64(INPUT_TORCH_SEED), "stride": np.int64(INPUT_STRIDE)}
    for name, t in (("visual_inputs", raw), ("crop_visual_inputs", crop), ("context_visual_inputs", ctx)):
        out = norm(t.float())
        g[name + "_sub"] = npf(out[..., ::INPUT_STRIDE, ::INPUT_STRIDE])
        g[name + "_sum"] = out.double().sum((-1, -2)).numpy()
        g[name + "_sqsum"] = (out.double() ** 2).sum((-1, -2)).numpy()
    unit = norm(raw.float() / 255.0)       # pixels already in 0..1: ImageNorm's data-dependent test must NOT rescale again
    g["unit_visual_inputs_sub"] = npf(unit[..., ::INPUT_STRIDE, ::INPUT_STRIDE])
    ids = det_caption_ids(6)
    torch.manual_seed(INPUT_TORCH_SEED)
    masked, labels = ns["mask_batch_text_tokens"](ids.clone(), _BertUncasedTokenizerStub())
    g["mlm_input_ids"], g["mlm_masked_ids"], g["mlm_labels"] = ids.numpy(), masked.numpy(), labels.numpy()
    np.savez_compressed(os.path.join(HERE, fname), **g)


def case_optimizer(fname):
    """a22 / N2 pinned to the reference: its own AdamW (src/optimization/adamw.py:40-103), get_lr_sched (src/optimization/sched.py:28) and
    torch's clip_grad_norm_, driven in the order of run_pretrain_sparse.py:615-646 (lr of the step -> clip -> step -> zero_grad) for three
    steps on closed-form parameters and gradients, in two scenarios (the release hyper-parameters; weight decay > 0 with a tighter clip)."""
    import warnings
    from torch.nn.utils import clip_grad_norm_
    from src.optimization.adamw import AdamW
    from src.optimization.sched import get_lr_sched
    g = {}
    for sc, hp in OPT_SCENARIOS.items():
        params = [torch.nn.Parameter(t.clone()) for t in opt_tensors("param")]
        opt = AdamW(params, lr=hp["lr"], betas=hp["betas"], weight_decay=hp["weight_decay"])
        for step in range(3):
            for p_, gr in zip(params, opt_tensors("grad", step)):
                p_.grad = gr.clone()
            lr_this_step = get_lr_sched(step + 1, hp["decay"], hp["lr"], hp["num_train_steps"], warmup_ratio=hp["warmup_ratio"])
            for pg in opt.param_groups:
                pg["lr"] = lr_this_step
            total = clip_grad_norm_(params, hp["grad_norm"])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")     # the deprecated add_(Number, Tensor) overloads of adamw.py:77-101 still run on this torch
                opt.step()
            opt.zero_grad()
            g["%s/lr/%d" % (sc, step)] = np.float64(lr_this_step)
            g["%s/grad_norm/%d" % (sc, step)] = np.float64(float(total))
            g["%s/params/%d" % (sc, step)] = np.concatenate([npf(p_).reshape(-1) for p_ in params])
        g[sc + "/exp_avg"] = np.concatenate([npf(opt.state[p_]["exp_avg"]).reshape(-1) for p_ in params])
        g[sc + "/exp_avg_sq"] = np.concatenate([npf(opt.state[p_]["exp_avg_sq"]).reshape(-1) for p_ in params])
    np.savez_compressed(os.path.join(HERE, fname), **g)


def main():
    am, _ = rh.import_reference()
    torch.set_num_threads(8)
    only = set(sys.argv[1:])  # e.g. `python -m tests.golden.make_golden prompter retrieval_grads` regenerates just those

    def want(name):
        return not only or name in only
    keys = {}
    if want("retrieval"):
        keys["retrieval_T2"] = case_retrieval(am, 2, 3, "retrieval_T2_B3.npz")
    if want("pretrain"):
        keys["pretrain_T8"] = case_pretrain(am, 8, 2, "pretrain_T8_B2.npz", with_grads=True)
    if want("block"):
        case_block_droppath("block11_droppath_T2_B4.npz")
    if want("prompter"):
        case_prompter(am, 2, 3, 8, "prompter_T2_B3_E8.npz")
    if want("retrieval_grads"):
        case_retrieval_grads(am, 2, 3, "retrieval_grads_T2_B3.npz")
    if want("retrieval_eval"):
        case_retrieval_eval(am, 2, 5, "retrieval_eval_T2_V5.npz")
    if want("retrieval_T16"):
        case_retrieval_frames(am, 16, 2, "retrieval_T16_B2.npz")
    if want("input_pipeline"):
        case_input_pipeline("input_pipeline_B4_T2.npz")
    if want("pretrain_release"):
        case_pretrain_release(am, "pretrain_release_T4_L30_B2.npz")
    if want("optimizer"):
        case_optimizer("optimizer_adamw_3steps.npz")
    if len(keys) == 2:
        json.dump(keys, open(os.path.join(HERE, "state_keys.json"), "w"), indent=0, sort_keys=True)
    for f in sorted(os.listdir(HERE)):
        if f.endswith((".npz", ".json")):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
