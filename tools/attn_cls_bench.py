"""Micro-benchmark (GPU): what the precise CLS parts and the log-sum-exp rows add to the spatial attention forward (alpro_attn_fwd, L = 197, fp16):
    python tools/attn_cls_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt, H, L, T = torch.float16, 12, 197, 8


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (64, 32, 64, 32):
    batch = B * T
    qkv = torch.randn(batch * L, 3 * H * 64, device="cuda").to(dt)
    cq = torch.randn(B, 3 * H * 64, device="cuda")
    co = torch.empty(batch, H * 64, device="cuda")
    row = []
    for cls in (False, True):
        for lse in (False, True):
            kw = dict(want_lse=lse)
            if cls:
                kw.update(cls_q=cq, cls_group=T, cls_out=co)
            row.append("%s%s %.1f us" % ("cls" if cls else "plain", " + lse" if lse else "", timeit(lambda: hip.attn(qkv, batch, L, H, 0.125, **kw))))
    print("B = %d: " % B + " | ".join(row))
