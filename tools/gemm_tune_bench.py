"""Interleaved A/B of the persistent GEMM's ALPRO_GEMM_TUNE variants (bf16, identity map, no activation):
    python tools/gemm_tune_bench.py [tune ...]     (default 0 1 2)
Each variant is checked against an fp32 matmul of the same operands first; timings are medians over interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip

hip.load()
hip.set_option("gemm_tile", int("256"))
dt = torch.bfloat16
M = 50208
shapes = [("qkv", M, 2304, 768), ("proj->bf16", M, 768, 768), ("fc1 nogelu", M, 3072, 768), ("fc2->bf16", M, 768, 3072),
          ("qkv B=64", 2 * M, 2304, 768), ("square4096", 4096, 4096, 4096), ("square8192", 8192, 8192, 8192)]
tunes = sys.argv[1:] or ["0", "1", "2"]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").to(dt)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    out = torch.empty(m, n, device="cuda", dtype=dt)
    ref = a[:2048].float() @ w.float().t()
    line = "%-12s M=%d N=%d K=%d " % (name, m, n, k)
    times = {t: [] for t in tunes}
    for t in tunes:
        hip.set_option("gemm_tune", int(t))
        out.zero_()
        hip.gemm(a, w, out=out)
        err = (out[:2048].float() - ref).abs().max().item() / ref.abs().max().item()
        tail = (out[-300:].float() - a[-300:].float() @ w.float().t()).abs().max().item() / ref.abs().max().item()
        assert err < 2e-2 and tail < 2e-2, (name, t, err, tail)
    for rnd in range(5):
        for t in tunes:
            hip.set_option("gemm_tune", int(t))
            hip.gemm(a, w, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                hip.gemm(a, w, out=out)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / 6)
    for t in tunes:
        ms = sorted(times[t])[len(times[t]) // 2]
        line += "| tune%s %.3f ms %4.0f TF (min %.3f) " % (t, ms, 2.0 * m * n * k / ms / 1e9, min(times[t]))
    print(line, flush=True)
