"""alpro_layernorm_bwd(_emit) in the three forms a ViT block's backward issues at B = 64 x 8 frames (norm2: identity rows + FRAME emit; norm1:
FRAME_TOKENS scatter + SKIP_CLS emit + bias column sums; temporal norm: SKIP_CLS scatter + ROWS emit incl. the CLS rows), fixed-order
reductions (workspace) vs fp32 atomics, workgroup count capped by the `ln_grid` option.   python tools/ln_bwd_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from alpro_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, N, D = 8, 196, 768
S = 1 + N * T
dev = torch.device("cuda", 0)
hip.load()
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, S, D, device=dev, generator=g)
gam = 1 + 0.1 * torch.randn(D, device=dev, generator=g)
dt = torch.float16
dy_frame = (torch.randn(B * T * (N + 1), D, device=dev, generator=g) * 0.1).to(dt)
dy_all = (torch.randn(B * S, D, device=dev, generator=g) * 0.1).to(dt)
dy_skip = (torch.randn(B * N * T, D, device=dev, generator=g) * 0.1).to(dt)
dx = torch.randn(B, S, D, device=dev, generator=g)
dg, db, cp = (torch.zeros(D, device=dev) for _ in range(3))
drop_s = torch.ones(B * T, device=dev)
drop_t = torch.ones(B * N, device=dev)
drop_m = torch.ones(B, device=dev)
forms = {
    "norm2  (identity rows, FRAME emit)": lambda: hip.layernorm_bwd(dy_all, x, gam, 1e-6, dx, dg, db, emit=dict(mode=hip.EMIT_FRAME, rows=B * T * (N + 1), dtype=dt, T=T, N=N, scale=drop_s)),
    "norm1  (FRAME_TOKENS, SKIP_CLS emit + colsum)": lambda: hip.layernorm_bwd(dy_frame, x, gam, 1e-6, dx, dg, db, rows=B * T * (N + 1), map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N,
                                                                              emit=dict(mode=hip.EMIT_SKIP_CLS, rows=B * N * T, dtype=dt, T=T, N=N, scale=drop_t, group=T, colsum_pre=cp)),
    "tnorm  (SKIP_CLS, ROWS emit + CLS rows)": lambda: hip.layernorm_bwd(dy_skip, x, gam, 1e-6, dx, dg, db, rows=B * N * T, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T,
                                                                        emit=dict(mode=hip.EMIT_ROWS, rows=B * S, dtype=dt, T=T, N=N, scale=drop_m, group=S, extra_cls=B)),
}


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in forms.items():
    cells = []
    for det in (True, False):
        hip.set_deterministic(det)
        for grid in ((0, 4096, 8192, 1024, 512) if det else (0,)):
            with hip.option("ln_grid", grid):
                cells.append("%s grid %-4s %6.1f us" % ("fixed-order" if det else "atomics    ", grid or "2048", timed(fn)))
    print("%-48s | %s" % (name, " | ".join(cells)))
hip.set_deterministic(True)
