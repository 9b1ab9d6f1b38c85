"""Sweep of the number of token ranges (tn_splits) of alpro_gemm_tn_acc at the ViT token count, with and without the atomic epilogue
(tn_kind 1 = ablation): python tools/gemm_tn_slices.py      (experiment 8 of profiles/r2_gemm_epilogue_experiments.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.bfloat16
M = 100416
def t(a, b, c, n=8):
    for _ in range(2): hip.gemm_tn_acc(a, b, c, atomic=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): hip.gemm_tn_acc(a, b, c, atomic=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, K in ((3072, 768), (2304, 768), (768, 768), (768, 3072)):
    a = torch.randn(M, N, device="cuda").to(dt); b = torch.randn(M, K, device="cuda").to(dt)
    c = torch.zeros(N, K, device="cuda")
    print("N=%d K=%d  tiles=%d   ranges: us with atomics / without" % (N, K, ((N + 255) // 256) * ((K + 255) // 256)))
    for s in (0, 8, 16, 24, 32, 40, 48, 56, 64, 72, 96, 128):
        r = []
        for kind in (0, 1):
            with hip.option("tn_splits", s), hip.option("tn_kind", kind):
                r.append(t(a, b, c))
        print("   %4s  %7.0f %7.0f" % (s or "auto", r[0], r[1]))
