"""The 8-phase GEMM's tile walk under CU theft, kernel level (round 5):  python tools/sched_contention.py [--dtype fp16]

For every shape of the table: time per launch (HIP events, 20 launches) with the static round-robin walk (gemm_sched 0) and with the dynamic
per-XCD ticket walk (gemm_sched 1), undisturbed and while 8 / 16 / 32 workgroups of tools/cu_thief.hip (64 KiB of LDS each: a collective's
channel kernels) stay resident on a second stream; results of every launch are compared bitwise with the undisturbed static-walk result.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from alpro_amd import hip  # noqa: E402
from tools.overlap_contention import build_thief  # noqa: E402

SHAPES = [("qkv B=64", 100416, 2304, 768, {}), ("proj B=64", 100416, 768, 768, {}), ("proj B=32", 50208, 768, 768, {}), ("qkv B=32", 50208, 2304, 768, {}),
          ("fc1+gelu' B=64", 100416, 3072, 768, {"save": True}), ("fc2 f32 res B=64", 100416, 768, 3072, {"f32": True}), ("fusion 180 tiles", 15168, 768, 768, {})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    hip.load()
    lib = build_thief()
    side = torch.cuda.Stream()
    buf = torch.zeros(64 * 65536, device="cuda")
    started = torch.zeros(1, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    cases = []
    for name, M, N, K, o in SHAPES:
        A = (torch.randn(M, K, device="cuda", generator=g)).to(dt)
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
        kw = dict(bias=torch.randn(N, device="cuda", generator=g))
        if o.get("save"):
            kw.update(act=hip.ACT_GELU_SAVE_GRAD, pre_act=torch.empty(M, N, dtype=dt, device="cuda"))
        if o.get("f32"):
            kw.update(residual=torch.randn(M, N, device="cuda", generator=g), out_dtype=torch.float32)
        out = torch.empty(M, N, dtype=kw.get("out_dtype", dt), device="cuda")
        with hip.option("gemm_sched", 0):
            ref = hip.gemm(A, W, **kw).clone()
        cases.append((name, M, N, K, A, W, kw, out, ref))
    torch.cuda.synchronize()
    # clocks settle over the first few hundred ms of load: the process' first measurements read 10-20 % slow otherwise (the "+6 % of the ticket
    # walk on qkv B=64" of this table's first version, profiles/r5_gemm_stagger_probe.txt)
    name, M, N, K, A, W, kw, out, ref = cases[0]
    for _ in range(600):
        hip.gemm(A, W, out=out, **kw)
    torch.cuda.synchronize()
    table = {}
    for stolen in (0, 8, 16, 32):
        started.zero_()
        t_launch = time.time()
        secs = 3.0
        if stolen:
            assert lib.cu_thief_launch(buf.data_ptr(), stolen, int(100e6 * secs), started.data_ptr(), side.cuda_stream) == 0
            while int(started.item()) < stolen:
                time.sleep(0.001)
        for name, M, N, K, A, W, kw, out, ref in cases:
            for sched in (0, 1, 0, 1):   # A / B / A / B, the faster of the two
                with hip.option("gemm_sched", sched):
                    for _ in range(3):
                        hip.gemm(A, W, out=out, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        hip.gemm(A, W, out=out, **kw)
                    e1.record()
                    e1.synchronize()
                    ok = torch.equal(out, ref)
                us = e0.elapsed_time(e1) / args.iters * 1e3
                prev = table.get((name, stolen, sched))
                if prev is None or us < prev[0] or not ok:
                    table[(name, stolen, sched)] = (us, 2.0 * M * N * K / us / 1e6, ok and (prev is None or prev[2]))
        still = (time.time() - t_launch) < secs
        torch.cuda.synchronize()
        if stolen and not still:
            print("WARNING: the thief ran out before the last measurement of stolen=%d" % stolen)
    print("%-18s %6s | %21s | %21s | %21s | %21s" % ("shape (%s)" % args.dtype, "walk", "undisturbed", "8 CUs taken", "16 CUs taken", "32 CUs taken"))
    for name, M, N, K, *_ in cases:
        for sched in (0, 1):
            base = table[(name, 0, sched)][0]
            cells = []
            for stolen in (0, 8, 16, 32):
                us, tf, ok = table[(name, stolen, sched)]
                cells.append("%7.1f us %5.0f TF %s%s" % (us, tf, "" if ok else "WRONG ", "" if stolen == 0 else "%+.0f%%" % (100 * (us / base - 1))))
            print("%-18s %6s | %s" % (name, "static" if sched == 0 else "ticket", " | ".join("%21s" % c for c in cells)))
    cost = [100 * (table[(n, 0, 1)][0] / table[(n, 0, 0)][0] - 1) for n, *_ in cases]
    print("undisturbed cost of the ticket walk per shape: " + ", ".join("%+.2f%%" % c for c in cost))


if __name__ == "__main__":
    main()
