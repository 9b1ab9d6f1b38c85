import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
hip.set_option("gemm_tile", int(sys.argv[1] if len(sys.argv) > 1 else "256"))
M, N = 50176, 2304
for out_f32 in (False, True):
    for K in (64, 128, 256, 768, 1536, 3072):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
        for _ in range(3):
            hip.gemm(a, w, out=out, out_dtype=out.dtype)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.gemm(a, w, out=out, out_dtype=out.dtype)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tiles = (M // 256) * (N // 256)
        rounds = tiles / 256.0
        print("out_f32=%d K=%5d  %.3f ms  %.0f TF  per-tile-round %.1f us" % (out_f32, K, ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / rounds))
