"""Micro-benchmark of the attention kernels on the model's shapes: python tools/attn_bench.py [fwd|bwd|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.float16 if os.environ.get("ALPRO_BENCH_DTYPE", "bf16") == "fp16" else torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"
print("dtype", dt)
H = 12
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# "head-major": the same 6144 (sequence, head) units as ONE-head sequences, i.e. rows of [q | k | v] = 384 bytes, a unit's 197 rows contiguous
# (75 KiB) instead of 128-byte pieces of 4608-byte rows -- same arithmetic, same bytes, only the layout differs (layout experiment, round 3)
for name, batch, L, bias, dp, H in [("vit spatial B=64", 512, 197, False, 0.0, 12), ("vit spatial B=32", 256, 197, False, 0.0, 12), ("bert text", 64, 40, True, 0.1, 12),
                                    ("fusion pos", 64, 237, True, 0.1, 12), ("fusion 4B", 256, 237, True, 0.1, 12), ("vit B=64 head-major", 512 * 12, 197, False, 0.0, 1),
                                    ("fusion 4B head-major", 256 * 12, 237, True, 0.1, 1)]:
    qkv = torch.randn(batch * L, 3 * H * 64, device="cuda").to(dt)
    kb = (torch.zeros(batch, L, device="cuda") if bias else None)
    fl = 4.0 * batch * H * L * L * 64
    if what in ("fwd", "all"):
        ms = timeit(lambda: hip.attn(qkv, batch, L, H, 0.125, key_bias=kb, want_lse=True, drop_p=dp, drop_seed=(123 if dp else 0)))
        print("%-16s fwd batch=%d L=%d: %.3f ms  %.1f TF/s  %.0f GB/s" % (name, batch, L, ms, fl / ms / 1e9, 4 * qkv.shape[0] * 768 * 2 / ms / 1e6))
    if what in ("bwd", "all"):
        out, lse = hip.attn(qkv, batch, L, H, 0.125, key_bias=kb, want_lse=True, drop_p=dp, drop_seed=(123 if dp else 0))
        do = torch.randn_like(out)
        ms = timeit(lambda: hip.attn_bwd(qkv, out, do, lse, batch, L, H, 0.125, key_bias=kb, drop_p=dp, drop_seed=(123 if dp else 0)))
        print("%-16s bwd batch=%d L=%d: %.3f ms  %.1f TF/s" % (name, batch, L, ms, 2.5 * fl / ms / 1e9))

H = 12
rows, T = 64 * 196 * 8, 8
qkv = torch.randn(rows, 3 * H * 64, device="cuda").to(dt)
if what in ("fwd", "all"):
    ms = timeit(lambda: hip.attn_temporal(qkv, T, H, 0.125, want_lse=True))
    print("vit temporal     fwd rows=%d T=%d: %.3f ms  %.0f GB/s" % (rows, T, ms, 4 * rows * 768 * 2 / ms / 1e6))
if what in ("bwd", "all"):
    out, lse = hip.attn_temporal(qkv, T, H, 0.125, want_lse=True)
    do = torch.randn_like(out)
    ms = timeit(lambda: hip.attn_temporal_bwd(qkv, out, do, lse, T, H, 0.125))
    print("vit temporal     bwd rows=%d T=%d: %.3f ms  %.0f GB/s (q,k,v,dO read + dq,dk,dv written)" % (rows, T, ms, 7 * rows * 768 * 2 / ms / 1e6))
