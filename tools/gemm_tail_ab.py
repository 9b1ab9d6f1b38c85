"""GPU A/B of the round-6 tail hand-over (option gemm_tail) on the B = 32 shapes whose last round of 256 x 256 tiles is nearly empty:
python tools/gemm_tail_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.float16


def timeit(fn, n=30):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K, kw in [("temporal proj B=32", 50176, 768, 768, {}), ("spatial proj B=32", 50432, 768, 768, {}), ("fc1 + GELU B=32", 50208, 3072, 768, dict(act=hip.ACT_GELU)),
                          ("fc2 fp32 residual B=32", 50208, 768, 3072, "res"), ("qkv B=32", 50176, 2304, 768, {}), ("temporal proj B=64", 100352, 768, 768, {}),
                          ("fc1 + GELU B=64", 100416, 3072, 768, dict(act=hip.ACT_GELU))]:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda")
    if kw == "res":
        res = torch.randn(M, N, device="cuda")
        kw = dict(out_dtype=torch.float32, residual=res)
    tiles = ((M + 255) // 256) * (N // 256)
    t = {}
    for rep in range(2):
        for tail in (0, 1):
            with hip.option("gemm_tail", tail):
                t.setdefault(tail, []).append(timeit(lambda: hip.gemm(a, w, bias=b, **kw)))
    fl = 2.0 * M * N * K
    print("%-24s %5d tiles = %.2f rounds | unsplit %.1f us (%.0f TF/s) | hand-over %.1f us (%.0f TF/s) | %+.1f %%" % (
        name, tiles, tiles / 256.0, min(t[0]), fl / min(t[0]) / 1e6, min(t[1]), fl / min(t[1]) / 1e6, (min(t[1]) / min(t[0]) - 1) * 100))
