"""GPU: how far ahead of the device does the host run in the B = 64 pretrain step?  Per step: host time until step() returns (everything queued),
and wall time until the device is done.  A host share near 100 % would mean the step is launch-bound, whatever the kernels do.
python tools/host_time_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from alpro_amd import config as rt, hip
from alpro_amd.modeling.alpro_models import AlproForPretrain
from alpro_amd.optim import FlatAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
hip.load()
rt.set_compute_dtype("fp16")
rt.set_cls_precise("auto")
torch.manual_seed(1234)
cfg = bench.Cfg(dict(bench.BERT_CFG, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1))
model = AlproForPretrain(cfg, dict(bench.VENC, num_frm=8)).to(dev).train()
opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.0, max_grad_norm=20.0)
batch = bench.synth_batch(B, 8, dev, seed=0, full=True)
marks = {}


def step():
    t = time.perf_counter()
    o = model(batch)
    marks["fwd"] = time.perf_counter() - t
    loss = o["mlm_loss"] + o["itm_loss"] + o["itc_loss"] + o["mpm_loss"]
    opt.backward(loss)
    marks["bwd"] = time.perf_counter() - t
    opt.step()
    opt.zero_grad()
    marks["all"] = time.perf_counter() - t


with torch.enable_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for it in range(6):
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("step %d: host %.1f ms (forward queued at %.1f, backward at %.1f), device done at %.1f ms -> host share %.0f %%"
              % (it, (t1 - t0) * 1e3, marks["fwd"] * 1e3, marks["bwd"] * 1e3, (t2 - t0) * 1e3, 100 * (t1 - t0) / (t2 - t0)))
    # back to back (the bench's regime): host never waits except where the step itself synchronises
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(6):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("6 steps back to back: host %.1f ms per step, device %.1f ms per step" % ((t1 - t0) / 6 * 1e3, (t2 - t0) / 6 * 1e3))
