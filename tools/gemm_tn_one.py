"""One weight-gradient shape for kernel traces / counters: python tools/gemm_tn_one.py M N K [atomic] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
M, N, K = (int(x) for x in sys.argv[1:4])
atomic = len(sys.argv) > 4 and sys.argv[4] == "1"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
a = torch.randn(M, N, device="cuda").to(torch.bfloat16)
b = torch.randn(M, K, device="cuda").to(torch.bfloat16)
c = torch.zeros(N, K, device="cuda")
cs = torch.zeros(N, device="cuda")
for _ in range(iters):
    hip.gemm_tn_acc(a, b, c, colsum=cs, atomic=atomic)
torch.cuda.synchronize()
