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
