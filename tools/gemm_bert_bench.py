"""128^2 vs persistent 256^2 kernel on the BERT-side shapes: python tools/gemm_bert_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.bfloat16
for M in (2560, 5120, 15168):
    for (N, K, res) in ((2304, 768, False), (768, 768, False), (768, 768, True), (3072, 768, False), (768, 3072, False), (768, 3072, True), (768, 2304, False)):
        a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        r = torch.randn(M, N, device="cuda") if res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if res else dt)
        line = "M=%5d N=%4d K=%4d res=%d " % (M, N, K, res)
        for tile in ("128", "256"):
            hip.set_option("gemm_tile", int(tile))
            for _ in range(3): hip.gemm(a, w, out=out, out_dtype=out.dtype, residual=r)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): hip.gemm(a, w, out=out, out_dtype=out.dtype, residual=r)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += "| tile%s %.1f us %4.0f TF " % (tile, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
        print(line)
