"""Text-video retrieval evaluation on cached encoder outputs.

Same result records and metrics as the reference's `inference_retrieval` / `eval_retrieval`
(src/tasks/run_video_retrieval.py:515-690), but every video goes through the visual encoder once and every caption through the
text encoder once; only the fusion pass + ITM head runs per (video, caption) pair, in mini-batches.  For V videos x C captions
with mini-batches of b captions the reference runs the ViT V*ceil(C/b) times and the text encoder on V*C captions.
"""
import math
from collections import defaultdict

import numpy as np
import torch


@torch.no_grad()
def inference_retrieval_cached(model, videos, text_input_ids, text_input_mask, caption_ids, eval_bsz=64, num_clips=1):
    """videos: iterable of (vid_id, visual_inputs (1, num_clips*num_frm, C, H, W)); the captions are shared by all videos
    (each reference batch carries one video and all captions).  Returns the reference's list of
    dict(vid_id, txt_id, score, sim): score = softmax(itm logits)[:, 1], sim = ITC similarity, both rounded to 4 decimals."""
    model.eval()
    n = text_input_ids.shape[0]
    text_cache = []
    for i in range(0, n, eval_bsz):
        ids, mask = text_input_ids[i:i + eval_bsz], text_input_mask[i:i + eval_bsz]
        emb, feat = model.encode_text(ids, mask)
        text_cache.append((emb, feat, mask, caption_ids[i:i + eval_bsz]))
    res = []
    for vid_id, visual_inputs in videos:
        T = visual_inputs.shape[1] // num_clips
        clips = visual_inputs.view((1, num_clips, T) + tuple(visual_inputs.shape[2:]))
        if num_clips != 1:
            raise NotImplementedError("the reference's score aggregation is only defined for inference_n_clips == 1 "
                                      "(run_video_retrieval.py:672-681 squeezes the clip axis)")
        ve, vf = model.encode_video(clips[:, 0])
        for emb, feat, mask, cids in text_cache:
            out = model.score_pairs(ve, vf, emb, feat, mask)
            probs = torch.softmax(out["logits"].float(), dim=1)[:, 1].tolist()
            sims = out["itc_scores"].float().reshape(-1).tolist()
            for cid, sc, sim in zip(cids, probs, sims):
                res.append(dict(vid_id=vid_id, txt_id=cid, score=round(sc, 4), sim=round(sim, 4)))
    return res


@torch.no_grad()
def score_all_pairs(model, videos, text_input_ids, text_input_mask, pair_bsz=512, text_bsz=1024):
    """Every caption against every video with nothing leaving the device: videos (V, T, C, H, W) or an iterable of (1, T, C, H, W)
    clips, captions (C, Lt).  Each video / caption is encoded ONCE; the V*C fusion passes run in flat mini-batches of `pair_bsz`
    (video, caption) pairs gathered from the two caches -- the GEMMs see M = pair_bsz * 237 rows whatever V and C are, where the
    reference loop (run_video_retrieval.py:642-690) runs one video x eval_bsz captions at a time.
    Returns (score (V, C) = softmax(itm_logits)[:, 1], sim (V, C) = ITC similarity), fp32 device tensors (unrounded)."""
    model.eval()
    ve, vf = [], []
    clips = videos if torch.is_tensor(videos) else torch.cat(list(videos), 0)
    for i in range(0, clips.shape[0], 8):
        e, f = model.encode_video(clips[i:i + 8])
        ve.append(e)
        vf.append(f)
    ve, vf = torch.cat(ve, 0), torch.cat(vf, 0)                                     # (V, 1+N, D), (V, 256)
    te, tf = [], []
    for i in range(0, text_input_ids.shape[0], text_bsz):
        e, f = model.encode_text(text_input_ids[i:i + text_bsz], text_input_mask[i:i + text_bsz])
        te.append(e)
        tf.append(f)
    te, tf = torch.cat(te, 0), torch.cat(tf, 0)                                     # (C, Lt, D), (C, 256)
    V, C = ve.shape[0], te.shape[0]
    sim = vf @ tf.t() / model.temp
    score = torch.empty(V * C, dtype=torch.float32, device=ve.device)
    ones = torch.ones(ve.shape[:2], dtype=text_input_mask.dtype, device=ve.device)
    for s in range(0, V * C, pair_bsz):
        idx = torch.arange(s, min(V * C, s + pair_bsz), device=ve.device)
        vi, ci = idx // C, idx % C
        out = model._fusion(torch.cat([te[ci], ve[vi]], dim=1), torch.cat([text_input_mask[ci], ones[vi]], dim=1))
        from alpro_amd.modeling.alpro_models import _linear32
        score[s:s + idx.numel()] = torch.softmax(_linear32(out[:, 0, :], model.itm_head).float(), dim=1)[:, 1]
    return score.view(V, C), sim


def records_from_matrices(score, sim, vid_ids, txt_ids):
    """(V, C) matrices -> the reference's result records (vid_id, txt_id, score, sim rounded to 4 decimals, :683-689), video-major."""
    sc, sm = score.detach().float().cpu().tolist(), sim.detach().float().cpu().tolist()
    return [dict(vid_id=v, txt_id=t, score=round(sc[i][j], 4), sim=round(sm[i][j], 4)) for i, v in enumerate(vid_ids) for j, t in enumerate(txt_ids)]


@torch.no_grad()
def retrieval_metrics_on_device(score, gt_col):
    """R@1/5/10, median and mean rank of a (rows, cols) device score matrix with ground-truth column gt_col[row], without sorting
    and without leaving the device until the five numbers are read: rank = 1 + #(columns scoring higher) + #(equal-scoring columns
    with a smaller index), i.e. the position a STABLE descending sort gives the ground truth (the reference sorts on the host,
    :541-555; on tie-free scores the two agree exactly)."""
    gt_col = gt_col.to(score.device).view(-1, 1)
    gt = score.gather(1, gt_col)
    cols = torch.arange(score.shape[1], device=score.device)[None, :]
    rank = 1 + (score > gt).sum(1) + ((score == gt) & (cols < gt_col)).sum(1)
    r = rank.double()
    out = torch.stack([(rank <= 1).double().mean() * 100, (rank <= 5).double().mean() * 100, (rank <= 10).double().mean() * 100, r.median() if r.numel() % 2 else
                       (r.sort()[0][r.numel() // 2 - 1] + r.sort()[0][r.numel() // 2]) / 2, r.mean()]).tolist()
    return dict(r1=out[0], r5=out[1], r10=out[2], medianR=out[3], meanR=out[4])


@torch.no_grad()
def topk_on_device(score, k):
    """Top-k columns of every row (values, indices) on the device -- e.g. the k best videos per caption of score.t()."""
    return torch.topk(score, min(k, score.shape[1]), dim=1)


def get_retrieval_metric_from_bool_matrix(bool_matrix):
    """R@1/5/10, median and mean rank from a (#rows, #cols) matrix sorted by decreasing score with exactly one ground-truth
    1 per row (run_video_retrieval.py:515-538)."""
    num_row = bool_matrix.shape[0]
    rows, gt_ranks = np.where(bool_matrix == 1)
    assert np.array_equal(rows, np.arange(num_row)), "each row should only a single GT"
    return dict(r1=100 * bool_matrix[:, 0].sum() / num_row, r5=100 * bool_matrix[:, :5].sum() / num_row,
                r10=100 * bool_matrix[:, :10].sum() / num_row, medianR=np.median(gt_ranks + 1), meanR=np.mean(gt_ranks + 1))


def _retrieval_scores(score_matrix, gt_row2col_id, row_idx2id, col_id2idx):
    order = torch.sort(score_matrix, dim=1, descending=True)[1]
    gt = torch.tensor([[col_id2idx[gt_row2col_id[row_idx2id[i]]]] for i in range(score_matrix.shape[0])])
    return get_retrieval_metric_from_bool_matrix((order == gt).numpy())


def eval_retrieval(vid_txt_score_dicts, gt_txt_id2vid_id):
    """text->video and video->text metrics from result records (run_video_retrieval.py:558-628; duplicates of a (txt, vid)
    pair keep the first record)."""
    by_txt = defaultdict(dict)
    for d in vid_txt_score_dicts:
        by_txt[d["txt_id"]].setdefault(d["vid_id"], d)
    txt_ids = list(by_txt)
    vid_ids = list(by_txt[txt_ids[0]])
    for t in txt_ids:
        assert len(by_txt[t]) == len(vid_ids), "each captions should be compared with the same #videos."
    txt_id2idx = {t: i for i, t in enumerate(txt_ids)}
    vid_id2idx = {v: i for i, v in enumerate(vid_ids)}
    score = torch.zeros(len(txt_ids), len(vid_ids))
    for t, preds in by_txt.items():
        for v, p in preds.items():
            score[txt_id2idx[t], vid_id2idx[v]] = p["score"]
    t2v = _retrieval_scores(score, gt_txt_id2vid_id, {i: t for t, i in txt_id2idx.items()}, vid_id2idx)
    gt_vid2txt = {v: k for k, v in gt_txt_id2vid_id.items()}
    v2t = _retrieval_scores(score.t(), gt_vid2txt, {i: v for v, i in vid_id2idx.items()}, txt_id2idx)
    return dict(text2video=t2v, video2text=v2t)
