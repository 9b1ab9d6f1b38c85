"""Micro-benchmark of alpro_gemm on the ViT shapes: python tools/gemm_bench.py [tile ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip

hip.load()
dt = torch.bfloat16
M = 50208
shapes = [("qkv", M, 2304, 768, {}), ("proj->bf16", M, 768, 768, {}), ("proj+res f32", M, 768, 768, {"res": True}),
          ("fc1 gelu", M, 3072, 768, {"act": hip.ACT_GELU}), ("fc1 nogelu", M, 3072, 768, {}), ("fc2+res f32", M, 768, 3072, {"res": True}),
          ("square4096", 4096, 4096, 4096, {}), ("square8192", 8192, 8192, 8192, {})]
tiles = sys.argv[1:] or ["128", "256"]
for name, m, n, k, opt in shapes:
    a = torch.randn(m, k, device="cuda").to(dt)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda")
    res = torch.randn(m, n, device="cuda") if opt.get("res") else None
    out = torch.empty(m, n, device="cuda", dtype=torch.float32 if res is not None else dt)
    line = "%-14s M=%d N=%d K=%d " % (name, m, n, k)
    for tile in tiles:
        hip.set_option("gemm_tile", int(tile))
        kw = dict(out=out, bias=bias, act=opt.get("act", 0), out_dtype=out.dtype, residual=res)
        for _ in range(3):
            hip.gemm(a, w, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.gemm(a, w, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += "| tile%s %.3f ms %.0f TF " % (tile, ms, 2.0 * m * n * k / ms / 1e9)
    print(line)
