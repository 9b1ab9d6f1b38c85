"""alpro_gemm_tn_acc over the wgrad shapes of the model, workspace mode (default), atomic mode, and without any epilogue (tn_kind 1, measurement only):
python tools/gemm_tn_hybrid.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
dt = torch.bfloat16
def t(a, b, c, n=10, atomic=False, cs=None):
    for _ in range(6): hip.gemm_tn_acc(a, b, c, colsum=cs, atomic=atomic)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): hip.gemm_tn_acc(a, b, c, colsum=cs, atomic=atomic)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (100416, 60672, 5120):
    for N, K in ((3072, 768), (768, 3072), (2304, 768), (768, 768), (30528, 768)):
        a = torch.randn(M, N, device="cuda").to(dt); b = torch.randn(M, K, device="cuda").to(dt)
        c = torch.zeros(N, K, device="cuda")
        r = []
        for kind, atomic in ((0, False), (0, False), (0, True), (1, True)):
            with hip.option("tn_kind", kind):
                r.append(t(a, b, c, atomic=atomic))
        csb = torch.zeros(N, device="cuda")
        r.append(t(a, b, c, atomic=None, cs=csb))
        ref = (a.float().t() @ b.float())
        c.zero_(); hip.gemm_tn_acc(a, b, c); torch.cuda.synchronize()
        err = ((c - ref).abs().max() / ref.abs().max()).item()
        print("M=%6d N=%4d K=%4d  workspace %6.0f %6.0f us (%4.0f TF)  atomics %6.0f  no epilogue %6.0f  default+bias grad %6.0f us  relerr %.1e" % (M, N, K, r[0], r[1], 2.0 * M * N * K / min(r[:2]) / 1e6, r[2], r[3], r[4], err))
