import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.bfloat16
M, N, K = 100416, 2304, 768
a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt); bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=dt)
def seg(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): hip.gemm(a, w, out=out, bias=bias)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("burst of 6 after idle:", ["%.0f" % (2.0*M*N*K/seg(6)/1e9) for _ in range(3)])
e = []
for i in range(10):
    e.append(seg(100))
print("sustained 10 x 100 launches (TF/s):", ["%.0f" % (2.0*M*N*K/x/1e9) for x in e])
