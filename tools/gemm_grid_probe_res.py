"""Per-tile time of the persistent GEMM on the fp32-residual shape with the grid capped to G workgroups: is the epilogue bound per CU
(per-tile time independent of G) or by the HBM share (per-tile time drops with fewer active CUs)?  python tools/gemm_grid_probe_res.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
hip.set_option("gemm_tile", 256)
dt = torch.bfloat16
for (M, N, K, res) in [(204800, 768, 768, True), (204800, 768, 768, False), (204800, 768, 128, True)]:
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    r = torch.randn(M, N, device="cuda") if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if res else dt)
    tiles = (M // 256) * (N // 256)
    for G in (256, 192, 128, 64, 32, 8):
        hip.set_option("gemm_grid", G)
        for _ in range(2): hip.gemm(a, w, out=out, residual=r, out_dtype=out.dtype)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): hip.gemm(a, w, out=out, residual=r, out_dtype=out.dtype)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        per = -(-tiles // G)
        print("M=%d N=%d K=%d res=%d grid %3d: %.3f ms, %d tiles/WG -> %.2f us per tile, %.0f TF, %.2f TB/s epilogue+A traffic" % (
            M, N, K, res, G, ms, per, ms * 1e3 / per, 2.0 * M * N * K / ms / 1e9, (M * N * (8 if res else 2) + M * K * 2) / ms / 1e9))
    hip.set_option("gemm_grid", 0)
