"""Where does the B=64 proxy's MLM / ITM error sit?  fp16 (no CLS side path) against the exact fp32 HIP mode, AlproForPretrain eval, B = 64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from alpro_amd import config as rt, hip  # noqa: E402
from alpro_amd.modeling.alpro_models import AlproForPretrain  # noqa: E402

hip.load()
torch.manual_seed(4)
B, T = 64, 8
m = AlproForPretrain(bench.Cfg(bench.BERT_CFG), dict(bench.VENC, num_frm=T)).eval().cuda()
batch = bench.synth_batch(B, T, "cuda", seed=11, full=True)
batch["text_input_mask"] = batch["text_input_mask"].clone()
batch["text_input_mask"][::3, 31:] = 0
torch.multinomial = lambda w, n=1, *a, **k: w.argmax(dim=-1, keepdim=True)
negs = {}
orig = AlproForPretrain._sample_negatives


def record(sim_v2t, sim_t2v, bs):
    if "n" not in negs:
        negs["n"] = orig(sim_v2t, sim_t2v, bs)
    return negs["n"]


AlproForPretrain._sample_negatives = staticmethod(record)


def run(dt):
    with rt.use_compute_dtype(dt), rt.use_cls_precise("0"), torch.no_grad():
        te = m._text_embeds(batch["mlm_text_input_ids"], batch["text_input_mask"])
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        va = torch.ones(ve.size()[:-1], dtype=torch.long, device="cuda")
        fo = m._fusion(torch.cat([te, ve], 1), torch.cat([batch["text_input_mask"], va], 1))
        out = m(batch)
    return dict(te=te.double().cpu(), ve=ve.double().cpu(), fo=fo.double().cpu(), mlm=out["mlm_scores"].double().cpu(), itm=out["itm_scores"].double().cpu())


ref, got = run("fp32"), run("fp16")
mask = batch["text_input_mask"].cpu().bool()
for k in ("te", "ve", "fo", "itm"):
    e = (got[k] - ref[k]).abs()
    print("%-4s max err %.3e  rms %.3e  ref rms %.3e" % (k, e.max(), e.pow(2).mean().sqrt(), ref[k].pow(2).mean().sqrt()))
e = (got["mlm"] - ref["mlm"]).abs()          # (B, 40, V)
print("mlm  max err %.3e rms %.3e ref rms %.3e ref max %.3e" % (e.max(), e.pow(2).mean().sqrt(), ref["mlm"].pow(2).mean().sqrt(), ref["mlm"].abs().max()))
pos = e.amax(-1)                              # (B, 40)
print("per-position max err: valid tokens max %.3e, padded tokens max %.3e" % (pos[mask].max(), pos[~mask].max() if (~mask).any() else 0.0))
b, t = divmod(int(pos.argmax()), 40)
print("worst position: caption %d token %d (valid=%s); its fusion-output error %.3e" % (b, t, bool(mask[b, t]), (got["fo"][b, t] - ref["fo"][b, t]).abs().max()))
fe = (got["fo"] - ref["fo"]).abs().amax(-1)   # (B, 237)
print("fusion output: text rows valid max %.3e, text rows padded max %.3e, video rows max %.3e" % (fe[:, :40][mask].max(), fe[:, :40][~mask].max(), fe[:, 40:].max()))
