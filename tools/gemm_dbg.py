import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
os.environ["ALPRO_GEMM_TILE"] = "256"
M, N = 50176, 2304
for K in (64, 768, 3072):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for alpha, name in ((1.0, "full (stagger)"), (-779.0, "full, no stagger")):
        for _ in range(3):
            hip.gemm(a, w, out=out, alpha=alpha)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.gemm(a, w, out=out, alpha=alpha)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("K=%d %-32s %.3f ms  per-tile-round %.1f us" % (K, name, ms, ms * 1e3 / (196 * 9 / 256.0)))
