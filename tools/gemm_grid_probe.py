"""Per-tile time of the persistent 256x256 GEMM with the grid capped to G workgroups (is the epilogue HBM-burst bound
or per-CU bound?): python tools/gemm_grid_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
hip.set_option("gemm_tile", int("256"))
dt = torch.bfloat16
for (M, N, K) in [(204800, 768, 768), (204800, 768, 3072), (204800, 768, 128)]:
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    tiles = (M // 256) * (N // 256)
    for G in (256, 128, 64, 32, 8):
        hip.set_option("gemm_grid", G)
        for _ in range(2): hip.gemm(a, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): hip.gemm(a, w, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        per = -(-tiles // G)
        print("M=%d N=%d K=%d grid %3d: %.3f ms, %d tiles/WG -> %.2f us per tile, %.0f TF" % (M, N, K, G, ms, per, ms * 1e3 / per, 2.0 * M * N * K / ms / 1e9))
