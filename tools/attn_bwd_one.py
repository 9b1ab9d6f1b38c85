"""A few launches of the spatial attention backward on the model's shape, for counter runs: python tools/attn_bwd_one.py [batch L bias dropout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
batch, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 197)
bias = len(sys.argv) > 3 and sys.argv[3] == "1"
dp = 0.1 if len(sys.argv) > 4 and sys.argv[4] == "1" else 0.0
dt = torch.float16 if os.environ.get("ALPRO_BENCH_DTYPE", "fp16") == "fp16" else torch.bfloat16
H = 12
qkv = torch.randn(batch * L, 3 * H * 64, device="cuda").to(dt)
kb = torch.zeros(batch, L, device="cuda") if bias else None
out, lse = hip.attn(qkv, batch, L, H, 0.125, key_bias=kb, want_lse=True, drop_p=dp, drop_seed=(123 if dp else 0))
do = torch.randn_like(out)
for _ in range(6):
    hip.attn_bwd(qkv, out, do, lse, batch, L, H, 0.125, key_bias=kb, drop_p=dp, drop_seed=(123 if dp else 0))
torch.cuda.synchronize()
