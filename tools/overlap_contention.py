"""CU contention between the persistent GEMMs and a collective's kernels, measured on ONE GPU.

    python tools/overlap_contention.py [--steps 4]        -> table of ms per training step over (stolen CUs) x (cu_budget)

A "CU thief" (tools/cu_thief.hip: n resident workgroups with an RCCL-channel-like footprint on a second stream) runs for the whole timed region
while the B = 64 pretraining step runs on the main stream; `cu_budget` is the library knob that sizes the persistent NT GEMM grids and the
weight-gradient range plan (alpro_amd.dist sets it while the overlapped gradient exchange is in flight).  Reading the table: the row with
budget 256 is what the round-3 code did under an overlapped exchange; the diagonal (budget = 256 - stolen) is the policy.
"""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alpro_amd import config as rt, hip  # noqa: E402


def build_thief():
    src, so = os.path.join(ROOT, "tools", "cu_thief.hip"), "/tmp/libcu_thief.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.cu_thief_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    hip.load()
    rt.set_compute_dtype("fp16")
    dev = torch.device("cuda", 0)
    from alpro_amd.modeling.alpro_models import AlproForPretrain
    from alpro_amd.optim import FlatAdamW
    torch.manual_seed(1234)
    model = AlproForPretrain(bench.Cfg(bench.BERT_CFG), dict(bench.VENC, num_frm=8)).to(dev).train()
    opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, max_grad_norm=20.0)
    batch = bench.synth_batch(args.batch, 8, dev, seed=0, full=True)

    def step():
        out = model(batch)
        opt.backward(out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"])
        opt.step()
        opt.zero_grad()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lib = build_thief()
    side = torch.cuda.Stream()
    buf = torch.zeros(64 * 65536, device=dev)
    started = torch.zeros(1, dtype=torch.int32, device=dev)
    rows = []
    for sched, stolen, budget in [(s_, st, bu) for s_ in (1, 0) for st in (0, 8, 16, 32) for bu in (256, 248, 240, 224)]:
        if True:
            if budget != 256 and (budget != 256 - stolen or sched == 0):   # the policy diagonal, for the ticket walk; the static walk's table is profiles/r4_overlap_cu_contention.txt
                continue
            hip.set_option("gemm_sched", sched)
            hip.set_option("cu_budget", 0 if budget == 256 else budget)
            step()                                       # (plans / grids of this budget warm)
            torch.cuda.synchronize()
            started.zero_()
            if stolen:
                ticks = int(100e6 * (0.25 * args.steps + 0.3))     # 100 MHz wall clock: a little longer than the timed region
                rc = lib.cu_thief_launch(buf.data_ptr(), stolen, ticks, started.data_ptr(), side.cuda_stream)
                assert rc == 0, rc
                while int(started.item()) < stolen:               # resident before the step's first kernel is launched
                    time.sleep(0.001)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.current_stream().synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            torch.cuda.synchronize()                              # (the thief runs out by itself)
            rows.append((sched, stolen, budget, ms))
            print("gemm_sched %d  stolen CUs %2d  cu_budget %3d  %.2f ms / step" % (sched, stolen, budget, ms), flush=True)
    hip.set_option("cu_budget", 0)
    hip.set_option("gemm_sched", 1)
    for sc in (1, 0):
        base = [ms for s_, s, b, ms in rows if s_ == sc and s == 0 and b == 256][0]
        print("\ngemm_sched %d (%s), relative to its undisturbed step (%.2f ms):" % (sc, "per-XCD ticket walk" if sc else "static round-robin walk", base))
        for s_, s, b, ms in rows:
            if s_ == sc:
                print("  stolen %2d budget %3d: %+.1f %%" % (s, b, 100.0 * (ms / base - 1.0)))


if __name__ == "__main__":
    main()
