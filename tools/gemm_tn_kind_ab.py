"""A/B of the weight-gradient GEMM schedules on the model's shapes (B = 64): round-3 one-barrier-per-stage loop (tn_kind 0) against the round-4
two-group schedule (tn_kind 2); interleaved rounds, random fp16 data, median.   python tools/gemm_tn_kind_ab.py [rounds]"""
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
shapes = [("qkv wgrad", M, 2304, 768), ("proj wgrad", M, 768, 768), ("fc1 wgrad", M, 3072, 768), ("fc2 wgrad", M, 768, 3072), ("fusion ffn wgrad", 60672, 3072, 768),
          ("fusion dense wgrad", 60672, 768, 768), ("text qkv wgrad", 5120, 2304, 768)]
for name, m, n, k in shapes:
    a = torch.randn(m, n, device="cuda").to(dt)
    b = torch.randn(m, k, device="cuda").to(dt)
    c = torch.zeros(n, k, device="cuda")
    cs = torch.zeros(n, device="cuda")
    ms = {0: [], 2: []}
    for _ in range(rounds):
        for kind in (0, 2):
            hip.set_option("tn_kind", kind)
            for _ in range(2):
                hip.gemm_tn_acc(a, b, c, colsum=cs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                hip.gemm_tn_acc(a, b, c, colsum=cs)
            e1.record()
            torch.cuda.synchronize()
            ms[kind].append(e0.elapsed_time(e1) / 6)
    hip.set_option("tn_kind", 0)
    fl = 2.0 * m * n * k / 1e9
    m0, m2 = statistics.median(ms[0]), statistics.median(ms[2])
    print("%-20s M=%6d N=%4d K=%4d | round-3 %.3f ms %5.0f TF | two-group %.3f ms %5.0f TF (%+.1f %%)" % (name, m, n, k, m0, fl / m0, m2, fl / m2, 100.0 * (m0 / m2 - 1.0)), flush=True)
