"""A/B of the NT GEMM kernels on the model's big shapes (B = 64 x 8 frames): round-3 persistent kernel (gemm_kind 0) against the round-4
8-phase two-group kernel (gemm_kind 1), and torch.matmul (hipBLASLt) as the vendor reference.  Interleaved rounds in ONE process, random fp16
data, median of the rounds (cdna_hip_programming.md 5.4 rules 24 / 25).

    python tools/gemm_kind_ab.py [rounds]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from alpro_amd import hip  # noqa: E402

hip.load()
dt = torch.float16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
M = 100352
shapes = [("qkv", M, 2304, 768, {}), ("proj 16-bit out", M, 768, 768, {}), ("fc1 gelu+save", M, 3072, 768, {"act": hip.ACT_GELU_SAVE_GRAD}),
          ("fc2 f32 residual", M, 768, 3072, {"res": True}), ("dgrad fc2 mul_saved", M, 3072, 768, {"act": hip.ACT_MUL_SAVED}), ("dgrad fc1", M, 768, 3072, {}),
          ("dgrad qkv", M, 768, 2304, {}), ("fusion dense", 60672, 768, 768, {}), ("fusion ffn1", 60672, 3072, 768, {"act": hip.ACT_GELU_SAVE_GRAD}),
          ("B=32 qkv", 50176, 2304, 768, {}), ("B=32 proj", 50176, 768, 768, {}), ("square 8192", 8192, 8192, 8192, {})]


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, m, n, k, opt in shapes:
    a = torch.randn(m, k, device="cuda").to(dt)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda")
    res = torch.randn(m, n, device="cuda") if opt.get("res") else None
    out = torch.empty(m, n, device="cuda", dtype=torch.float32 if res is not None else dt)
    c2 = torch.randn(m, n, device="cuda").to(dt) if opt.get("act") else None
    kw = dict(out=out, bias=bias, act=opt.get("act", 0), out_dtype=out.dtype, residual=res, pre_act=c2)
    wt = w.t().contiguous()
    ms = {0: [], 1: [], "vendor": []}
    for _ in range(rounds):
        for kind in (0, 1):
            with hip.option("gemm_kind", kind):
                ms[kind].append(timed(lambda: hip.gemm(a, w, **kw)))
        if not opt:
            ms["vendor"].append(timed(lambda: torch.matmul(a, wt)))
    fl = 2.0 * m * n * k / 1e9
    med = {k_: (statistics.median(v) if v else None) for k_, v in ms.items()}
    print("%-22s M=%6d N=%4d K=%4d | round-3 %.3f ms %5.0f TF | 8-phase %.3f ms %5.0f TF (%+.1f %%) | %s" % (
        name, m, n, k, med[0], fl / med[0], med[1], fl / med[1], 100.0 * (med[0] / med[1] - 1.0),
        ("hipBLASLt %.3f ms %5.0f TF" % (med["vendor"], fl / med["vendor"])) if med["vendor"] else ""), flush=True)
