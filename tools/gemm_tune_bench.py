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
shapes = [("qkv", M, 2304, 768, {}), ("proj->bf16", M, 768, 768, {}), ("fc1 nogelu", M, 3072, 768, {}), ("fc1 gelu+pre", M, 3072, 768, {"act": hip.ACT_GELU, "pre": 1}),
          ("fc1-dgrad gelu'", M, 768, 3072, {"act": hip.ACT_GELU_BWD, "pre": 2}), ("fc2->bf16", M, 768, 3072, {}),
          ("proj+res f32", M, 768, 768, {"res": True}), ("fc2+res f32", M, 768, 3072, {"res": True}),
          ("qkv B=64", 2 * M, 2304, 768, {}), ("proj+res B=64", 2 * M, 768, 768, {"res": True}), ("square4096", 4096, 4096, 4096, {}), ("square8192", 8192, 8192, 8192, {})]
tunes = sys.argv[1:] or ["0", "1", "2"]
for name, m, n, k, opt in shapes:
    a = torch.randn(m, k, device="cuda").to(dt)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    res = torch.randn(m, n, device="cuda") if opt.get("res") else None
    bias = torch.randn(n, device="cuda")
    pre = torch.empty(m, n, device="cuda", dtype=dt) if opt.get("pre") == 1 else (torch.randn(m, n, device="cuda").to(dt) if opt.get("pre") == 2 else None)
    out = torch.empty(m, n, device="cuda", dtype=torch.float32 if res is not None else dt)
    kw = dict(out=out, bias=bias, act=opt.get("act", 0), out_dtype=out.dtype, residual=res, pre_act=pre)
    lin = a[:2048].float() @ w.float().t() + bias
    if opt.get("act") == hip.ACT_GELU:
        ref = torch.nn.functional.gelu(lin)
    elif opt.get("act") == hip.ACT_GELU_BWD:
        p64 = pre[:2048].float().requires_grad_(True)
        torch.nn.functional.gelu(p64).sum().backward()
        ref = lin * p64.grad
    else:
        ref = lin
    if res is not None:
        ref = ref + res[:2048]
    line = "%-12s M=%d N=%d K=%d " % (name, m, n, k)
    times = {t: [] for t in tunes}
    for t in tunes:
        hip.set_option("gemm_tune", int(t))
        out.zero_()
        hip.gemm(a, w, **kw)
        err = (out[:2048].float() - ref).abs().max().item() / ref.abs().max().item()
        assert int(t) in (3, 4, 6, 10, 11, 12) or err < 2e-2, (name, t, err)   # 3 / 4 / 6 are ablations (no stores / no epilogue)
        if opt.get("pre") == 1 and int(t) not in (3, 4, 6, 10, 11, 12):
            assert ((pre[:2048].float() - lin).abs().max().item() / lin.abs().max().item()) < 2e-2, (name, t, "pre-activation copy")
    for rnd in range(5):
        for t in tunes:
            hip.set_option("gemm_tune", int(t))
            hip.gemm(a, w, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                hip.gemm(a, w, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / 6)
    for t in tunes:
        ms = sorted(times[t])[len(times[t]) // 2]
        line += "| tune%s %.3f ms %4.0f TF (min %.3f) " % (t, ms, 2.0 * m * n * k / ms / 1e9, min(times[t]))
    print(line, flush=True)
