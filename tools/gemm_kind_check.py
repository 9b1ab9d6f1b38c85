"""Diagnosis aid: the 8-phase NT GEMM (gemm_kind 1) and the round-3 kernel (gemm_kind 0) against an fp32 matmul of the same fp16 operands, on the
model's full-tile shapes and epilogues; prints WHERE an output differs (tile row / tile column / 64-column wave block / 16-row fragment row), and
repeats each case to expose timing-dependent results.   python tools/gemm_kind_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from alpro_amd import hip  # noqa: E402

hip.load()
dt = torch.float16
cases = [(10240, 1024, 3072, "f32res"), (10240, 1024, 3072, "plain"), (60672, 768, 3072, "plain"), (60672, 768, 768, "plain"), (60672, 2304, 768, "plain"),
         (60672, 3072, 768, "gelu"), (100352, 768, 3072, "f32res"), (100352, 768, 3072, "plain"), (20480, 512, 768, "f32res"), (10240, 1024, 1536, "f32res")]
for M, N, K, case in cases:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda")
    ref = a.float() @ w.float().t() + b
    kw = dict(bias=b)
    if case == "f32res":
        res = torch.randn(M, N, device="cuda")
        kw.update(residual=res, out_dtype=torch.float32)
        ref = ref + res
    if case == "gelu":
        kw["act"] = hip.ACT_GELU
        ref = torch.nn.functional.gelu(ref)
    tol = 2e-3 if case == "f32res" else 3e-2
    for kind in (0, 1):
        for rep in range(3 if kind == 1 else 1):
            with hip.option("gemm_kind", kind):
                out = hip.gemm(a, w, **kw).float()
            torch.cuda.synchronize()
            bad = (out - ref).abs() > tol * (1.0 + ref.abs())
            nbad = int(bad.sum())
            line = "M=%6d N=%4d K=%4d %-7s kind %d rep %d: max err %.3e, %d bad" % (M, N, K, case, kind, rep, float((out - ref).abs().max()), nbad)
            if nbad:
                rows, cols = bad.any(1).nonzero().flatten(), bad.any(0).nonzero().flatten()
                line += " | bad rows %d (tile rows %s, rows mod 256 in %s..%s, mod 16 set %s) | bad cols %d (tile cols %s, col blocks of 64: %s, mod 16 set %s)" % (
                    rows.numel(), sorted(set((rows // 256).tolist()))[:8], int((rows % 256).min()), int((rows % 256).max()), sorted(set((rows % 16).tolist())),
                    cols.numel(), sorted(set((cols // 256).tolist())), sorted(set(((cols % 256) // 64).tolist())), sorted(set((cols % 16).tolist())))
            print(line, flush=True)
