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
