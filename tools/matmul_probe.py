"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this model's shapes, next to alpro_gemm on the same operands:
a reference point for how far the hand-written NT kernel is from what the hardware allows.  python tools/matmul_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
for dt in (torch.bfloat16, torch.float16):
    for (M, N, K) in [(100352, 2304, 768), (100352, 768, 768), (100416, 3072, 768), (100416, 768, 3072), (50176, 2304, 768), (50176, 768, 768), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        res = []
        for name, fn in (("torch.matmul", lambda: torch.matmul(a, w.t())), ("alpro_gemm", lambda: hip.gemm(a, w))):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res.append("%s %.3f ms %.0f TF/s" % (name, ms, 2.0 * M * N * K / ms / 1e9))
        print("%s M=%d N=%d K=%d: %s" % (str(dt).replace("torch.", ""), M, N, K, " | ".join(res)), flush=True)
