import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = 256 * 1024 * 1024
a = torch.randn(N, device="cuda"); b = torch.empty_like(a)
ms = t(lambda: b.copy_(a)); print("torch copy 1GiB f32: %.3f ms  %.2f TB/s (r+w)" % (ms, 2 * N * 4 / ms / 1e9))
ms = t(lambda: b.zero_()); print("torch memset 1GiB: %.3f ms  %.2f TB/s (w)" % (ms, N * 4 / ms / 1e9))
ms = t(lambda: hip.cast(a, torch.bfloat16)); print("alpro cast f32->bf16 1GiB in: %.3f ms  %.2f TB/s (r+w)" % (ms, N * 6 / ms / 1e9))
ms = t(lambda: a.sum()); print("torch sum 1GiB: %.3f ms  %.2f TB/s (r)" % (ms, N * 4 / ms / 1e9))
x = torch.randn(50176, 768, device="cuda")
g = torch.ones(768, device="cuda"); be = torch.zeros(768, device="cuda")
ms = t(lambda: hip.layernorm(x, g, be, 1e-6, torch.bfloat16)); print("LN 50176x768 f32->bf16: %.3f ms  %.2f TB/s" % (ms, 50176 * 768 * 6 / ms / 1e9))
