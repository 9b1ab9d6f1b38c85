"""GPU: where does the B-pair pretrain step synchronise the host with the device?  torch.cuda.set_sync_debug_mode("warn") on one warmed step.
python tools/sync_probe.py [B]"""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from alpro_amd import config as rt, hip
from alpro_amd.modeling.alpro_models import AlproForPretrain
from alpro_amd.optim import FlatAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
hip.load()
rt.set_compute_dtype("fp16")
rt.set_cls_precise("auto")
torch.manual_seed(1234)
cfg = bench.Cfg(dict(bench.BERT_CFG, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1))
model = AlproForPretrain(cfg, dict(bench.VENC, num_frm=8)).to(dev).train()
opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.0, max_grad_norm=20.0)
batch = bench.synth_batch(B, 8, dev, seed=0, full=True)


def step():
    o = model(batch)
    loss = o["mlm_loss"] + o["itm_loss"] + o["itc_loss"] + o["mpm_loss"]
    opt.backward(loss)
    opt.step()
    opt.zero_grad()


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message).lower():
        st = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "sync_probe" not in f.filename]
        print("SYNC:", str(message)[:90], "|", " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in st[-4:][::-1]))


with torch.enable_grad():
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    step()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
print("done")
