"""A/B of gemm_tune values on the model's long shapes (bf16, plain epilogue -- the only instantiation that carries the variants):
    ALPRO_HIP_LIB=alpro_amd/lib/libalpro_hip_ablate.so python tools/gemm_tune_ab.py 1 5 6
Values 5 / 6 (s_setprio 1 / 3 around the MFMAs of each K sub-step) exist only in the measurement build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpro_amd import hip
hip.load()
tunes = [int(x) for x in sys.argv[1:]] or [1, 5, 6]
shapes = [(100864, 768, 768), (100864, 2304, 768), (100864, 3072, 768), (100864, 768, 3072)]
res = {}
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = None
    for rep in range(2):             # two interleaved passes: clock / thermal drift shows up as a difference between them
        for t in tunes:
            hip.set_option("gemm_tune", t)
            for _ in range(3):
                hip.gemm(a, w, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.gemm(a, w, out=out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            res.setdefault((M, N, K, t), []).append(ms)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), "gemm_tune %d changes the result" % t
    print("M=%d N=%d K=%d: " % (M, N, K) + "  ".join("tune %d: %s ms (%.0f TF/s)" % (t, "/".join("%.3f" % x for x in res[(M, N, K, t)]), 2.0 * M * N * K / min(res[(M, N, K, t)]) / 1e9) for t in tunes))
hip.set_option("gemm_tune", 1)
