"""Micro-benchmark (GPU) of two small once-per-step launches (round 6: tproj_small 152 -> 13-25 us with eight rows per trip; a one-sweep softmax_xent
with the row in registers -- 369 VGPRs, one wave per SIMD -- measured 259 us against 275-283: the kernel is bound by its expf calls, not by its three
sweeps, and was not kept): alpro_softmax_xent on the MLM head's shape (2560 x 30522 fp32 logits -> fp16 gradient) and
alpro_tproj_small mode 1 (the merged temporal projection's product rule, four blocks per launch).   python tools/small_kernels_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, V = 2560, 30522
logits = torch.randn(M, V, device="cuda") * 2
labels = torch.randint(0, V, (M,), device="cuda")
labels[::3] = -100
gs = torch.tensor([1.0 / M], device="cuda")
print("softmax_xent %d x %d -> fp16 gradient: %.1f us" % (M, V, timeit(lambda: hip.softmax_xent(logits, labels, grad_dtype=torch.float16, grad_scale=gs))))
D = 768
jobs = [dict(wfc=torch.randn(D, D, device="cuda"), bp=torch.randn(D, device="cuda"), db1=torch.randn(D, device="cuda"), g_fc=torch.zeros(D, D, device="cuda"),
             g_bp=torch.zeros(D, device="cuda")) for _ in range(4)]
table, n = hip.tproj_jobs(jobs, torch.device("cuda"))
print("tproj_small mode 1, 4 jobs: %.1f us" % timeit(lambda: hip.tproj_small(table, n, D, 1)))
