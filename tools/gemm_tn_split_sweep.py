"""Slice-count sweep of alpro_gemm_tn_acc on small token counts: python tools/gemm_tn_split_sweep.py  (relaunches itself per setting)"""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from alpro_amd import hip
    hip.load()
    dt = torch.bfloat16
    out = []
    for M in (5120, 15168):
        for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
            a = torch.randn(M, N, device="cuda").to(dt); b = torch.randn(M, K, device="cuda").to(dt)
            c = torch.zeros(N, K, device="cuda")
            for _ in range(5): hip.gemm_tn_acc(a, b, c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): hip.gemm_tn_acc(a, b, c)
            e1.record(); torch.cuda.synchronize()
            out.append("%.0f" % (e0.elapsed_time(e1) / 20 * 1e3))
    print(sys.argv[1].rjust(5), " ".join(x.rjust(6) for x in out))
else:
    print("splits  M=5120: 2304x768 768x768 3072x768 768x3072 | M=15168: same  (us)")
    for s in ("auto", "1", "2", "4", "8", "16", "24"):
        env = dict(os.environ)
        if s != "auto":
            env["ALPRO_TN_SPLITS"] = s
        subprocess.run([sys.executable, __file__, s], env=env)
