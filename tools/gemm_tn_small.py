"""Fixed cost of alpro_gemm_tn_acc on small token counts (BERT shapes): python tools/gemm_tn_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.bfloat16
for M in (64, 256, 1024, 2560, 15168):
    for N, K in ((768, 768), (3072, 768)):
        a = torch.randn(M, N, device="cuda").to(dt); b = torch.randn(M, K, device="cuda").to(dt)
        c = torch.zeros(N, K, device="cuda")
        for _ in range(3): hip.gemm_tn_acc(a, b, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.gemm_tn_acc(a, b, c)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("M=%6d N=%4d K=%4d  %.1f us  %.0f TF  splits=%s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, os.environ.get("ALPRO_TN_SPLITS", "auto")))
