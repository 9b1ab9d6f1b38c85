"""One GEMM shape for counter collection: python tools/gemm_one.py M N K [tile] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
M, N, K = (int(x) for x in sys.argv[1:4])
if len(sys.argv) > 4:
    hip.set_option("gemm_tile", int(sys.argv[4]))
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(iters):
    hip.gemm(a, w, out=out)
torch.cuda.synchronize()
