"""Micro-benchmark (GPU): the fused temporal half (alpro_gemm_qkv_tattn) against the two launches it replaces (alpro_gemm qkv + alpro_attn_temporal_fwd)
on the model's shapes: python tools/tattn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.float16 if os.environ.get("ALPRO_BENCH_DTYPE", "fp16") == "fp16" else torch.bfloat16


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


H, T, K = 12, 8, 768
for B in (32, 64):
    M = B * 196 * T
    a = (torch.randn(M, K, device="cuda") * 1.0).to(dt)
    w = (torch.randn(3 * H * 64, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(3 * H * 64, device="cuda")
    fl = 2.0 * M * 3 * H * 64 * K
    t_g = timeit(lambda: hip.gemm(a, w, bias=b))
    qkv = hip.gemm(a, w, bias=b)
    t_a = timeit(lambda: hip.attn_temporal(qkv, T, H, 0.125))
    t_f = timeit(lambda: hip.gemm_qkv_tattn(a, w, b, T, H, 0.125))
    t_t = timeit(lambda: hip.gemm_qkv_tattn(a, w, b, T, H, 0.125, want_qkv=True))
    t_a2 = timeit(lambda: hip.attn_temporal(qkv, T, H, 0.125, want_lse=True))
    print("B=%d training form: fused + q|k|v + lse written %.1f us against %.1f us (gemm + attention with lse)" % (B, t_t * 1e3, (t_g + t_a2) * 1e3))
    d = (hip.gemm_qkv_tattn(a, w, b, T, H, 0.125).float() - hip.attn_temporal(qkv, T, H, 0.125).float()).abs().max().item()
    print("B=%d M=%d: qkv gemm %.1f us (%.0f TF/s) + temporal attention %.1f us = %.1f us | fused %.1f us (%.0f TF/s on the GEMM's flops) | max |fused - two launches| %.2e" % (
        B, M, t_g * 1e3, fl / t_g / 1e9, t_a * 1e3, (t_g + t_a) * 1e3, t_f * 1e3, fl / t_f / 1e9, d))
