"""Packed 16-bit epilogue of the 8-phase GEMM against the staged fp32 one (option gemm_epi 1 / 0):  python tools/gemm_epi_ab.py [--dtype fp16]

Per shape: time per launch (HIP events, 20 launches, A/B/A/B after a warm-up) and whether the two epilogues give the same bits.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from alpro_amd import hip  # noqa: E402

SHAPES = [("qkv B=64", 100416, 2304, 768, {}), ("qkv B=32", 50208, 2304, 768, {}), ("proj-like B=64", 100416, 768, 768, {}),
          ("fc1 gelu+gelu' B=64", 100416, 3072, 768, {"act": "save"}), ("fc1 gelu B=32", 50208, 3072, 768, {"act": "gelu"}),
          ("fc1 dgrad B=64", 100416, 768, 3072, {"nobias": True}), ("qkv dgrad B=64", 100416, 768, 2304, {"nobias": True}),
          ("fusion 180 tiles", 15168, 768, 768, {}), ("ragged M=1000*16", 16000, 768, 768, {}),
          ("fc2 f32 res B=64 (unaffected)", 100416, 768, 3072, {"f32": True}), ("fc2 dgrad x gelu' (unaffected)", 100416, 3072, 768, {"mul": True})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    args = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    hip.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    print("%-30s | %9s %9s | %9s %9s | %7s | same bits" % ("shape (%s)" % args.dtype, "staged", "packed", "staged", "packed", "gain"))
    for name, M, N, K, o in SHAPES:
        A = (torch.randn(M, K, device="cuda", generator=g)).to(dt)
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
        kw = {} if o.get("nobias") else dict(bias=torch.randn(N, device="cuda", generator=g))
        pre = None
        if o.get("act") == "save":
            pre = torch.empty(M, N, dtype=dt, device="cuda")
            kw.update(act=hip.ACT_GELU_SAVE_GRAD, pre_act=pre)
        if o.get("act") == "gelu":
            kw.update(act=hip.ACT_GELU)
        if o.get("mul"):
            kw = dict(act=hip.ACT_MUL_SAVED, pre_act=torch.rand(M, N, device="cuda", generator=g).to(dt))
        if o.get("f32"):
            kw.update(residual=torch.randn(M, N, device="cuda", generator=g), out_dtype=torch.float32)
        out = torch.empty(M, N, dtype=kw.get("out_dtype", dt), device="cuda")
        res = {}
        for _ in range(10):
            hip.gemm(A, W, out=out, **kw)
        cells = []
        for epi in (0, 1, 0, 1):
            with hip.option("gemm_epi", epi):
                for _ in range(3):
                    hip.gemm(A, W, out=out, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    hip.gemm(A, W, out=out, **kw)
                e1.record()
                e1.synchronize()
                res[epi] = (out.clone(), None if pre is None else pre.clone())
            cells.append(e0.elapsed_time(e1) / 20 * 1e3)
        same = torch.equal(res[0][0], res[1][0]) and (pre is None or o.get("mul") or torch.equal(res[0][1], res[1][1]))
        tf = 2.0 * M * N * K / min(cells[1], cells[3]) / 1e6
        print("%-30s | %9.1f %9.1f | %9.1f %9.1f | %+6.1f%% | %s   (%.0f TF/s packed)" % (name, *cells, 100.0 * ((cells[1] + cells[3]) / (cells[0] + cells[2]) - 1.0), same, tf), flush=True)
    tiled_pair(dt)


def tiled_pair(dt):
    """fc1 forward (gelu + gelu') and fc2 dgrad (x gelu') at B = 64 with the saved factor in rows against the tile layout"""
    M, N, K = 100416, 3072, 768
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).to(dt)
    W1 = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    DY = torch.randn(M, K, device="cuda", generator=g).to(dt)
    W2T = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    b1 = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, dtype=dt, device="cuda")
    rows = hip.gemm_c2_tiled_rows(M, N, K, dt)
    bufs = {False: torch.empty(M, N, dtype=dt, device="cuda"), True: torch.empty(max(rows, 1), N, dtype=dt, device="cuda")}
    res = {}
    print("\nsaved gelu' in rows / in the tile layout (us per launch; %d buffer rows)" % rows)
    for what in ("fc1 gelu+gelu' B=64", "fc2 dgrad x gelu' B=64"):
        cells = []
        for tiled in (False, True, False, True):
            if tiled and not rows:
                cells.append(float("nan"))
                continue
            def run():
                if what.startswith("fc1"):
                    hip.gemm(A, W1, out=out, bias=b1, act=hip.ACT_GELU_SAVE_GRAD, pre_act=bufs[tiled], c2_tiled=tiled)
                else:
                    hip.gemm(DY, W2T, out=out, act=hip.ACT_MUL_SAVED, pre_act=bufs[tiled], c2_tiled=tiled)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            e1.synchronize()
            res[(what, tiled)] = out.clone()
            cells.append(e0.elapsed_time(e1) / 20 * 1e3)
        same = rows and torch.equal(res[(what, False)], res[(what, True)])
        print("%-30s | rows %8.1f tiled %8.1f | rows %8.1f tiled %8.1f | %+6.1f%% | same bits %s   (%.0f TF/s tiled)" % (
            what, *cells, 100.0 * ((cells[1] + cells[3]) / (cells[0] + cells[2]) - 1.0), same, 2.0 * M * N * K / min(cells[1], cells[3]) / 1e6), flush=True)


if __name__ == "__main__":
    main()
