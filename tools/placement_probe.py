"""Placement probe (GPU): does a kernel's time depend on WHERE its operand / output buffers sit?  (Round 6: the fused temporal kernel ran 177 us or
190 us inside the same forward depending on which buffers the caching allocator handed it, profiles/r6_split_streams_ab.txt.)
Carves the input and the output of a launch out of one arena at controlled byte offsets from a 2 MiB-aligned base and times the launch:
    python tools/placement_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.float16
H, T, K, B = 12, 8, 768, 32
M = B * 196 * T


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


arena = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
base = (-arena.data_ptr()) % (2 << 20)


def carve(off, rows, cols, dtype=dt):
    nb = rows * cols * (2 if dtype == dt else 4)
    return arena[base + off: base + off + nb].view(dtype).view(rows, cols)


w = (torch.randn(3 * H * 64, K, device="cuda") * 0.05).to(dt)
wp = (torch.randn(768, K, device="cuda") * 0.05).to(dt)
bias = torch.randn(3 * H * 64, device="cuda")
src = (torch.randn(M, K, device="cuda")).to(dt)
MB = 1 << 20
A_SPAN = 128 * MB    # the input lives in [0, 128 MiB), the output in [256 MiB, ...): offsets below are added to each
offs = [0, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 * MB, 1 * MB + 4096]
print("fused temporal kernel (alpro_gemm_qkv_tattn), B = 32: us per launch; rows = input offset, columns = output offset (bytes from a 2 MiB boundary)")
print("%10s" % "" + "".join("%9d" % o for o in offs))
for oa in offs:
    a = carve(oa, M, K)
    a.copy_(src)
    row = []
    for oo in offs:
        out = carve(256 * MB + oo, M, H * 64)
        row.append(timeit(lambda: hip.gemm_qkv_tattn(a, w, bias, T, H, 0.125, out=out)))
    print("%10d" % oa + "".join("%9.1f" % v for v in row))
print("8-phase qkv GEMM (M = 50176, N = 2304, K = 768): same table")
print("%10s" % "" + "".join("%9d" % o for o in offs))
for oa in offs[::2]:
    a = carve(oa, M, K)
    a.copy_(src)
    row = []
    for oo in offs:
        out = carve(256 * MB + oo, M, 3 * H * 64)
        row.append(timeit(lambda: hip.gemm(a, w, bias=bias, out=out)))
    print("%10d" % oa + "".join("%9.1f" % v for v in row))
print("8-phase projection GEMM (M = 50176, N = 768, K = 768): same table")
print("%10s" % "" + "".join("%9d" % o for o in offs))
for oa in offs[::2]:
    a = carve(oa, M, K)
    a.copy_(src)
    row = []
    for oo in offs:
        out = carve(256 * MB + oo, M, 768)
        row.append(timeit(lambda: hip.gemm(a, wp, bias=bias[:768], out=out)))
    print("%10d" % oa + "".join("%9.1f" % v for v in row))
# distance between input and output (both 2 MiB aligned): does it matter how far apart they are?
print("fused temporal kernel: output at input + d (d in MiB, both 2 MiB aligned)")
a = carve(0, M, K)
a.copy_(src)
for d in (74, 76, 78, 80, 96, 128, 130, 192, 256, 258, 384, 512, 640):
    out = carve(d * MB, M, H * 64)
    print("  d = %4d MiB: %.1f us" % (d, timeit(lambda: hip.gemm_qkv_tattn(a, w, bias, T, H, 0.125, out=out))))
