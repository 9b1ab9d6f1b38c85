"""Generate golden vectors by running the REFERENCE itself (CPU, fp32, eval) in this container.

    python -m tests.golden.make_golden            # writes tests/golden/*.npz + state_keys.json

The reference has no tests/fixtures of its own, so these files are the
pin for oracle/alpro_oracle.py.  Weights and inputs are NOT stored: both sides regenerate
them from oracle/det_init.py closed forms.  Stochastic ops are pinned:
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

from oracle.det_init import det_batch, fill_state_dict_, unit_uniform  # noqa: E402
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
    from oracle.det_init import det_param
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


def main():
    am, _ = rh.import_reference()
    torch.set_num_threads(8)
    keys = {}
    keys["retrieval_T2"] = case_retrieval(am, 2, 3, "retrieval_T2_B3.npz")
    keys["pretrain_T8"] = case_pretrain(am, 8, 2, "pretrain_T8_B2.npz", with_grads=True)
    case_block_droppath("block11_droppath_T2_B4.npz")
    json.dump(keys, open(os.path.join(HERE, "state_keys.json"), "w"), indent=0, sort_keys=True)
    for f in sorted(os.listdir(HERE)):
        if f.endswith((".npz", ".json")):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
