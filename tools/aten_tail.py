"""Where does the ATen tail of one training step come from?  One profiled step of bench.py's pretrain_step workload; every
device kernel that is not an alpro:: kernel is attributed to the innermost alpro_amd/ (or bench.py) Python frame that
launched it.    python tools/aten_tail.py [B]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from alpro_amd import config as rt, hip
from alpro_amd.modeling.alpro_models import AlproForPretrain
from alpro_amd.optim import FlatAdamW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hip.load()
rt.set_compute_dtype(os.environ.get("ALPRO_BENCH_DTYPE", "fp16"))
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = AlproForPretrain(bench.Cfg(bench.BERT_CFG), dict(bench.VENC, num_frm=8)).to(dev).train()
opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.0, max_grad_norm=20.0)
batch = bench.synth_batch(B, 8, dev, seed=0, full=True)


def step():
    out = model(batch)
    opt.backward(out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"])   # (scaled under fp16 operands)
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
try:
    xcfg = dict(experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True))   # (newer torch: python stacks need it)
except Exception:  # noqa: BLE001
    xcfg = {}
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, **xcfg) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    names = [k.name for k in e.kernels]
    if all("alpro::" in n for n in names):
        continue
    src = "?"
    for fr in e.stack:
        if ("alpro_amd/" in fr or "bench.py" in fr or "tools/aten_tail" in fr) and "alpro_amd/hip.py" not in fr:
            src = fr[fr.index("alpro_amd/"):] if "alpro_amd/" in fr else fr[-90:]
            break
    a = agg[(e.name, src)]
    a[0] += 1
    a[1] += sum(k.duration for k in e.kernels)
tot = sum(v[1] for v in agg.values())
print("non-alpro device time of one step: %.1f us in %d launches" % (tot, sum(v[0] for v in agg.values())))
for (op, src), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%8.1f us  n=%4d  %-28s %s" % (us, n, op[:28], src))
