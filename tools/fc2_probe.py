import sys, torch
sys.path.insert(0, "/root/repo")
from alpro_amd import hip
hip.load()
dt = torch.float16
M, N, K = 100416, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).to(dt)
W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
b = torch.randn(N, device="cuda", generator=g)
res = torch.randn(M, N, device="cuda", generator=g)
rs = torch.rand(64, device="cuda", generator=g)
def t(kw, od):
    out = torch.empty(M, N, dtype=od, device="cuda")
    for _ in range(5): hip.gemm(A, W, out=out, out_dtype=od, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.gemm(A, W, out=out, out_dtype=od, **kw)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for rep in range(2):
    print("f16 out, bias            %.1f us" % t(dict(bias=b), dt))
    print("f32 out, bias            %.1f us" % t(dict(bias=b), torch.float32))
    print("f32 out, bias + residual %.1f us" % t(dict(bias=b, residual=res), torch.float32))
    print("f32 out, bias + residual + row scale %.1f us" % t(dict(bias=b, residual=res, row_scale=rs, row_scale_group=1569), torch.float32))
    print("f16 out, bias + residual %.1f us" % t(dict(bias=b, residual=res), dt))
