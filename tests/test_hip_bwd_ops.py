"""GPU: backward kernels (C ABI) against torch autograd in fp64 on identical (pre-rounded) operands."""
import math

import pytest
import torch

from tests.test_hip_ops import DTYPES, _hip, close, rnd

pytestmark = pytest.mark.gpu

GRAD_TOL = {torch.float32: (2e-4, 2e-5), torch.bfloat16: (3e-2, 3e-2), torch.float16: (5e-3, 5e-3)}


def attn_ref(qkv, batch, L, H, scale, bias=None, group=None):
    t = qkv.view(batch, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (t[0] @ t[1].transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias[:, None, None, :].double()
    if group is not None:
        idx = torch.arange(L) // group
        s = s.masked_fill(idx[:, None] != idx[None, :], float("-inf"))
    return (s.softmax(-1) @ t[2]).transpose(1, 2).reshape(batch * L, H * 64)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("batch,L,masked", [(2, 40, True), (2, 197, False), (1, 237, True), (1, 70, False), (1, 130, False), (2, 161, True),
                                            (1, 192, False), (1, 225, False), (2, 256, True)])
def test_attn_bwd(dt, batch, L, masked):
    hip = _hip()
    H = 12
    qkv = (rnd(batch * L, 3 * H * 64, seed=70 + L) * 0.7).to(dt)
    dout = rnd(batch * L, H * 64, seed=71 + L).to(dt)
    bias = None
    if masked:
        m = torch.ones(batch, L)
        for b in range(batch):
            m[b, L - 4 - 3 * b:] = 0
        bias = (1.0 - m) * -10000.0
    q64 = qkv.double().requires_grad_(True)
    ref = attn_ref(q64, batch, L, H, 0.125, bias)
    ref.backward(dout.double())
    out, lse = hip.attn(qkv.cuda(), batch, L, H, 0.125, None if bias is None else bias.cuda(), want_lse=True)
    dqkv = hip.attn_bwd(qkv.cuda(), out, dout.cuda(), lse, batch, L, H, 0.125, None if bias is None else bias.cuda())
    g = q64.grad.view(batch * L, 3, H * 64)
    d = dqkv.view(batch * L, 3, H * 64)
    for i, name in enumerate("QKV"):
        close(d[:, i], g[:, i], *GRAD_TOL[dt], "d%s" % name)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,groups", [(8, 9), (4, 17), (2, 48), (16, 3)])
def test_attn_temporal_bwd(dt, T, groups):
    hip = _hip()
    H = 12
    rows = groups * T
    qkv = (rnd(rows, 3 * H * 64, seed=80 + T) * 0.7).to(dt)
    dout = rnd(rows, H * 64, seed=81 + T).to(dt)
    q64 = qkv.double().requires_grad_(True)
    attn_ref(q64, groups, T, H, 0.125).backward(dout.double())
    out, lse = hip.attn_temporal(qkv.cuda(), T, H, 0.125, want_lse=True)
    dqkv = hip.attn_temporal_bwd(qkv.cuda(), out, dout.cuda(), lse, T, H, 0.125)
    close(dqkv, q64.grad, *GRAD_TOL[dt], "temporal dqkv")


@pytest.mark.parametrize("dt", DTYPES)
def test_layernorm_bwd_maps(dt):
    hip = _hip()
    B, T, N, D = 2, 4, 9, 768
    S = 1 + N * T
    x = rnd(B, S, D, seed=90) * 2 + 0.3
    g, b = 1 + 0.1 * rnd(D, seed=91), 0.1 * rnd(D, seed=92)
    for mode, rows, kw in (("identity", B * S, {}), ("skip", B * N * T, dict(map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)),
                           ("frame", B * T * (N + 1), dict(map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N))):
        dy = rnd(rows, D, seed=93).to(dt)
        x64 = x.double().requires_grad_(True)
        g64, b64 = g.double().requires_grad_(True), b.double().requires_grad_(True)
        ln = torch.nn.functional.layer_norm(x64, (D,), g64, b64, 1e-6)
        if mode == "identity":
            y = ln.view(-1, D)
        elif mode == "skip":
            y = ln[:, 1:].reshape(-1, D)
        else:
            xs = ln[:, 1:].reshape(B, N, T, D).permute(0, 2, 1, 3)
            y = torch.cat([ln[:, :1].unsqueeze(1).expand(B, T, 1, D), xs], 2).reshape(-1, D)
        dres = rnd(B, S, D, seed=94)
        (y * dy.double()).sum().backward()
        dx = dres.clone().cuda()
        dg, db = torch.zeros(D).cuda(), torch.zeros(D).cuda()
        hip.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), 1e-6, dx, dg, db, rows=rows, **kw)
        close(dx, dres.double() + x64.grad, 1e-4, 2e-4, "ln bwd dx " + mode)
        close(dg, g64.grad, 1e-4, 2e-3, "ln bwd dgamma " + mode)
        close(db, b64.grad, 1e-4, 2e-3, "ln bwd dbeta " + mode)


def test_layernorm_bwd_two_streams():
    hip = _hip()
    rows, D = 37, 768
    x, dy, dy2 = rnd(rows, D, seed=95), rnd(rows, D, seed=96).to(torch.bfloat16), rnd(rows, D, seed=97)
    g = 1 + 0.1 * rnd(D, seed=98)
    x64 = x.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(x64, (D,), g.double(), torch.zeros(D, dtype=torch.float64), 1e-12)
    (y * (dy.double() + dy2.double())).sum().backward()
    dx = torch.empty(rows, D).cuda()
    hip.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), 1e-12, dx, torch.zeros(D).cuda(), torch.zeros(D).cuda(), dy2=dy2.cuda(), accumulate=False)
    close(dx, x64.grad, 1e-4, 2e-4, "ln bwd two streams")


@pytest.mark.parametrize("dt", DTYPES)
def test_transpose_gelu_misc(dt):
    hip = _hip()
    x = rnd(300, 200, seed=100).to(dt)
    cs = torch.ones(200).cuda()
    t = hip.transpose(x.cuda(), colsum=cs)
    assert t.shape == (200, 320)
    assert torch.equal(t[:, :300].cpu(), x.t().contiguous()) and float(t[:, 300:].abs().sum()) == 0
    close(cs, 1 + x.double().sum(0), 1e-5, 1e-4, "colsum")
    with _determinism(hip, False):   # the kernel's fused column sums (fp32 atomics per 64-row tile) instead of the fixed-order default
        cs = torch.ones(200).cuda()
        hip.transpose(x.cuda(), colsum=cs)
        close(cs, 1 + x.double().sum(0), 1e-5, 1e-4, "colsum (fused)")
    w = rnd(100, 768, seed=101)
    t2 = hip.transpose(w.cuda(), out_dtype=dt, pad_to=64)
    assert torch.equal(t2[:, :100].cpu(), w.to(dt).t().contiguous())
    u, dh = rnd(64, 3072, seed=102).to(dt), rnd(64, 3072, seed=103).to(dt)
    u64 = u.double().requires_grad_(True)
    (torch.nn.functional.gelu(u64) * dh.double()).sum().backward()
    close(hip.gelu_bwd(dh.cuda(), u.cuda()), u64.grad, *{torch.float32: (1e-5, 1e-5), torch.bfloat16: (1e-2, 1e-2), torch.float16: (2e-3, 2e-3)}[dt], "gelu bwd")
    if dt == torch.float32:
        dxo = rnd(3, 5, 768, seed=104)
        ds = hip.cls_mean_bwd(dxo.cuda(), 3, 4)
        close(ds, (dxo[:, 0] / 4).repeat_interleave(4, 0), 1e-6, 1e-6, "cls mean bwd")
        src, idx = rnd(50, 768, seed=105), torch.randint(0, 7, (50,))
        dst = torch.zeros(7, 768).cuda()
        hip.scatter_add_rows(src.cuda(), idx.cuda(), dst)
        close(dst, torch.zeros(7, 768, dtype=torch.float64).index_add_(0, idx, src.double()), 1e-5, 1e-5, "scatter add")
        dst = torch.zeros(10, 768).cuda()
        hip.scatter_add_rows(src.cuda(), None, dst, idx_mod=10)
        close(dst, src.double().view(5, 10, 768).sum(0), 1e-5, 1e-5, "scatter add mod")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_pre_activation_copy(dt):
    hip = _hip()
    a, w, b = rnd(300, 768, seed=110), rnd(256, 768, seed=111, scale=0.05), rnd(256, seed=112)
    pre = torch.empty(300, 256, dtype=dt).cuda()
    out = hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), bias=b.cuda(), act=hip.ACT_GELU, pre_act=pre)
    ref = a.to(dt).double() @ w.to(dt).double().T + b.double()
    tol = (2e-5, 2e-5) if dt == torch.float32 else (1e-2, 1e-2)
    close(pre, ref, *tol, "pre-activation")
    close(out, torch.nn.functional.gelu(ref), *tol, "gelu out")


def test_flat_adamw_matches_reference_update():
    """FlatAdamW (alpro_sumsq + alpro_adamw_step) vs the reference's HF-style AdamW + clip_grad_norm_ restated in torch fp64
    (src/optimization/adamw.py:77-101, run_pretrain_sparse.py:633), three steps, odd-sized tensors."""
    _hip()
    from alpro_amd.optim import FlatAdamW
    torch.manual_seed(0)
    shapes = [(7, 13), (768,), (5, 3, 2), (1,)]
    params = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    ref_p = [p.detach().double().cpu().clone() for p in params]
    ref_m = [torch.zeros_like(p) for p in ref_p]
    ref_v = [torch.zeros_like(p) for p in ref_p]
    opt = FlatAdamW(params, lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=2.0)
    for step in range(1, 4):
        grads = [torch.randn(*s) * 3 for s in shapes]
        for p, g in zip(params, grads):
            if p.grad is None:
                p.grad = g.cuda().clone()
            else:
                p.grad.copy_(g)
        opt.step()
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
        coef = min(2.0 / (float(total) + 1e-6), 1.0)
        for i, g in enumerate(grads):
            gg = g.double() * coef
            ref_m[i] = ref_m[i] * 0.9 + 0.1 * gg
            ref_v[i] = ref_v[i] * 0.98 + 0.02 * gg * gg
            ss = 1e-2 * math.sqrt(1 - 0.98 ** step) / (1 - 0.9 ** step)
            ref_p[i] = ref_p[i] - ss * ref_m[i] / (ref_v[i].sqrt() + 1e-6)
            ref_p[i] = ref_p[i] - 1e-2 * 0.01 * ref_p[i]
        for p, r in zip(params, ref_p):
            close(p, r, 1e-5, 1e-6, "adamw step %d" % step)
        opt.zero_grad()
        assert all(float(p.grad.abs().sum()) == 0 for p in params)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_adamw_step_refreshes_the_16_bit_mirror(dt):
    """Round 6 (alpro_adamw_step_lp): the optimizer pass also writes the 16-bit mirror of the parameters the GEMM operands are views of -- bit for
    bit what alpro_cast_from_f32 of the updated parameters gives (it replaced that launch), parameters / moments bit for bit those of the pass
    without a mirror, on a ragged size (two float4 per thread and iteration, the tail element by element), and a skipped step (non-finite
    squared norm under dynamic loss scaling) leaves parameters, moments AND mirror alone while still clearing the gradients."""
    hip = _hip()
    n = 256 * 4 * 37 + 4 * 5 + 3
    g = torch.Generator().manual_seed(5)
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    m0, v0 = torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.1
    norm = (g0.double() ** 2).sum().float().reshape(1).cuda()
    res = {}
    for with_lp in (False, True):
        p, gg, m, v = p0.clone().cuda(), g0.clone().cuda(), m0.clone().cuda(), v0.clone().cuda()
        lp = torch.full((n,), 7.0, dtype=dt).cuda() if with_lp else None
        hip.adamw_step(p, gg, m, v, 1e-2, 0.9, 0.98, 1e-6, 0.01, 1e-2, norm, 2.0, 1.0, zero_grad=True, lp=lp)
        res[with_lp] = (p, m, v, gg, lp)
    for a, b, what in zip(res[True][:4], res[False][:4], ("p", "m", "v", "g")):
        assert torch.equal(a, b), what
    assert float(res[True][3].abs().sum()) == 0
    assert torch.equal(res[True][4], hip.cast(res[True][0], dt)), "mirror != cast of the updated parameters"
    ref_p = p0.double()
    coef = min(2.0 / (math.sqrt(float((g0.double() ** 2).sum())) + 1e-6), 1.0)
    gr = g0.double() * coef
    rm, rv = m0.double() * 0.9 + 0.1 * gr, v0.double() * 0.98 + 0.02 * gr * gr
    ref_p = ref_p - 1e-2 * rm / (rv.sqrt() + 1e-6)
    ref_p = ref_p - 1e-2 * 0.01 * ref_p
    close(res[True][0], ref_p, 1e-5, 1e-6, "adamw with mirror")
    # overflow-skipped step
    p, gg, m, v = p0.clone().cuda(), g0.clone().cuda(), m0.clone().cuda(), v0.clone().cuda()
    lp = hip.cast(p, dt)
    lp_before = lp.clone()
    dyn = torch.tensor([65536.0, 0.0, 3.0, 0.0]).cuda()
    hip.adamw_step(p, gg, m, v, 1e-2, 0.9, 0.98, 1e-6, 0.01, 1e-2, torch.tensor([float("inf")]).cuda(), 2.0, 1.0, dyn_state=dyn, zero_grad=True, lp=lp)
    assert torch.equal(p.cpu(), p0) and torch.equal(m.cpu(), m0) and torch.equal(v.cpu(), v0) and torch.equal(lp, lp_before) and float(gg.abs().sum()) == 0


@pytest.mark.parametrize("tn_kind", [0, 2])
@pytest.mark.parametrize("atomic", [False, True])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(3138, 768, 768), (1000, 2304, 768), (6400, 768, 3072), (130, 30522, 768), (64, 8, 8), (4001, 520, 264), (50176, 768, 768)])
def test_gemm_tn_acc(dt, M, N, K, atomic, tn_kind):
    """Weight-gradient GEMM on natural layouts (tr-read operands, split over token ranges; partial tiles combined through the
    workspace + fixed-order reduce, or by fp32 atomics); tn_kind 2 = the two-group schedule of round 4 (same operands, same epilogue)."""
    hip = _hip()
    hip.set_option("tn_kind", tn_kind)
    ldn = (N + 7) // 8 * 8
    a = torch.zeros(M, ldn)
    a[:, :N] = rnd(M, N, seed=120) * 0.5
    b = rnd(M, K, seed=121)
    c0 = rnd(N, K, seed=122)
    c = c0.clone().cuda()
    bg = torch.full((ldn,), 2.0).cuda()
    hip.gemm_tn_acc(a.to(dt).cuda()[:, :N], b.to(dt).cuda(), c, colsum=bg, atomic=atomic)
    hip.set_option("tn_kind", hip._OPTION_DEFAULTS["tn_kind"])
    ref = c0.double() + a[:, :N].to(dt).double().T @ b.to(dt).double()
    close(c, ref, 2e-5, 2e-3 * math.sqrt(M / 1000.0), "gemm_tn_acc")
    close(bg[:N], 2 + a[:, :N].to(dt).double().sum(0), 1e-5, 1e-3, "gemm_tn_acc fused bias gradient")
    assert float((bg[N:] - 2).abs().max()) == 0 if ldn > N else True
    cs = torch.ones(ldn).cuda()
    hip.colsum_acc(a.to(dt).cuda(), cs)
    close(cs[:N], 1 + a[:, :N].to(dt).double().sum(0), 1e-5, 1e-3, "colsum_acc")


def _keep_mask(seed, n, p):
    """numpy restatement of common.hpp::drop_keep for indices 0..n-1."""
    import numpy as np
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = (np.uint32(seed) ^ (idx.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ ((idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x85EBCA77))).astype(np.uint32)
        h ^= h >> np.uint32(16); h *= np.uint32(0x7FEB352D); h ^= h >> np.uint32(15); h *= np.uint32(0x846CA68B); h ^= h >> np.uint32(16)
    return torch.from_numpy(((h >> np.uint32(8)) >= np.uint32(int(p * 16777216.0 + 0.5))).astype("float32"))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_dropout_gemm_and_backward_mask(dt):
    """Fused hidden dropout (xbert.py:358,436): GEMM epilogue mask == gather_cast mask == the hash restated in numpy."""
    hip = _hip()
    M, N, K, p, seed = 520, 768, 768, 0.1, 12345
    a, w, b, r = rnd(M, K, seed=130), rnd(N, K, seed=131, scale=0.05), rnd(N, seed=132), rnd(M, N, seed=133)
    keep = _keep_mask(seed, M * N, p).view(M, N)
    assert 0.88 < float(keep.mean()) < 0.92
    out = hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), bias=b.cuda(), out_dtype=torch.float32, residual=r.cuda(), drop_p=p, drop_seed=seed)
    ref = r.double() + keep.double() / (1 - p) * (a.to(dt).double() @ w.to(dt).double().T + b.double())
    close(out, ref, 2e-5, 3e-4, "dropout gemm f32 out")
    if dt != torch.float32:
        out16 = hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), bias=b.cuda(), drop_p=p, drop_seed=seed)
        close(out16, keep.double() / (1 - p) * (a.to(dt).double() @ w.to(dt).double().T + b.double()), 1e-2, 1e-2, "dropout gemm 16-bit out")
    g = rnd(M, N, seed=134)
    gc = hip.gather_cast(g.cuda(), torch.float32, drop_p=p, drop_seed=seed)
    close(gc, g.double() * keep.double() / (1 - p), 1e-6, 1e-6, "gather_cast dropout mask")


@pytest.mark.parametrize("L", [70, 237])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_attention_dropout_fwd_bwd(dt, L):
    """Attention-probability dropout (xbert.py:331) forward and backward against autograd with the same mask (L = 237: the fusion
    encoder's length, served by the key-owned 16-bit backward)."""
    hip = _hip()
    batch, H, p, seed = 2, 12, 0.1, 777
    qkv = (rnd(batch * L, 3 * H * 64, seed=140) * 0.7).to(dt)
    dout = rnd(batch * L, H * 64, seed=141).to(dt)
    keep = _keep_mask(seed, batch * H * L * L, p).view(batch, H, L, L).double()
    q64 = qkv.double().requires_grad_(True)
    t = q64.view(batch, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    pr = ((t[0] @ t[1].transpose(-1, -2)) * 0.125).softmax(-1) * keep / (1 - p)
    ref = (pr @ t[2]).transpose(1, 2).reshape(batch * L, H * 64)
    ref.backward(dout.double())
    out, lse = hip.attn(qkv.cuda(), batch, L, H, 0.125, want_lse=True, drop_p=p, drop_seed=seed)
    tol = {torch.float32: (2e-5, 2e-5), torch.bfloat16: (2e-2, 2e-2), torch.float16: (3e-3, 3e-3)}[dt]
    close(out, ref, *tol, "attn dropout fwd")
    dqkv = hip.attn_bwd(qkv.cuda(), out, dout.cuda(), lse, batch, L, H, 0.125, drop_p=p, drop_seed=seed)
    close(dqkv, q64.grad, *GRAD_TOL[dt], "attn dropout bwd")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind", [2])   # (3 / 4, the persistent variant, live in the measurement build since round 4)
@pytest.mark.parametrize("batch,L,masked,p", [(3, 197, False, 0.0), (2, 237, True, 0.1), (2, 140, False, 0.0), (1, 256, True, 0.1), (30, 197, False, 0.0),
                                              (64, 224, False, 0.0), (22, 193, False, 0.0)])
def test_attn_bwd_key_owned_vs_two_phase(dt, kind, batch, L, masked, p):
    """The key-owned backward (option attn_bwd = 2: every query-tile x key-tile pair once, dS handed to the dQ contraction through LDS;
    attn_bwd = 3 / 4: its persistent form for 7 key tiles without bias / dropout, with / without L2 touches, units walked by one workgroup
    per CU with the next query tiles in flight -- 30 x 12 and 64 x 12 units exercise 1, 2 and 3 units per workgroup; the default,
    attn_bwd = 1, picks per shape and is what test_attn_bwd runs) against the two-phase kernel (attn_bwd = 0)
    on the same inputs: dK and dV run the same MFMA sequence on the same operands -> bitwise equal; dQ differs only by the fp32
    summation order over the key tiles; and two runs are bitwise equal (no atomics)."""
    hip = _hip()
    H, seed = 12, (4242 if p else 0)
    qkv = (rnd(batch * L, 3 * H * 64, seed=310 + L) * 0.7).to(dt).cuda()
    dout = rnd(batch * L, H * 64, seed=311 + L).to(dt).cuda()
    bias = None
    if masked:
        bias = torch.zeros(batch, L)
        bias[:, L - 9:] = -10000.0
        bias = bias.cuda()
    out, lse = hip.attn(qkv, batch, L, H, 0.125, bias, want_lse=True, drop_p=p, drop_seed=seed)
    with hip.option("attn_bwd", 0):
        two = hip.attn_bwd(qkv, out, dout, lse, batch, L, H, 0.125, bias, drop_p=p, drop_seed=seed).view(batch * L, 3, H * 64)
    with hip.option("attn_bwd", kind):
        one = hip.attn_bwd(qkv, out, dout, lse, batch, L, H, 0.125, bias, drop_p=p, drop_seed=seed).view(batch * L, 3, H * 64)
        again = hip.attn_bwd(qkv, out, dout, lse, batch, L, H, 0.125, bias, drop_p=p, drop_seed=seed).view(batch * L, 3, H * 64)
    assert torch.equal(one, again)
    assert torch.equal(one[:, 1], two[:, 1]) and torch.equal(one[:, 2], two[:, 2])
    close(one[:, 0], two[:, 0].double().cpu(), *((2e-2, 2e-2) if dt == torch.bfloat16 else (3e-3, 3e-3)), "dQ")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,L,masked,p", [(16, 197, False, 0.0), (8, 237, True, 0.1), (24, 40, True, 0.1)])
def test_attention_unit_order_is_result_neutral(dt, batch, L, masked, p):
    """Option attn_order = 1 hands whole sequences (all 12 heads) to one XCD instead of spreading the heads of a sequence over the 8 XCDs:
    a permutation of which workgroup computes which (sequence, head) unit -- forward output, log-sum-exp and every gradient bitwise equal
    to the identity order (batch a multiple of 8: the permutation is active)."""
    hip = _hip()
    H, seed = 12, (99 if p else 0)
    qkv = (rnd(batch * L, 3 * H * 64, seed=410 + L) * 0.7).to(dt).cuda()
    dout = rnd(batch * L, H * 64, seed=411 + L).to(dt).cuda()
    bias = None
    if masked:
        bias = torch.zeros(batch, L)
        bias[:, L - 5:] = -10000.0
        bias = bias.cuda()
    res = []
    for order in (0, 1):
        with hip.option("attn_order", order):
            out, lse = hip.attn(qkv, batch, L, H, 0.125, bias, want_lse=True, drop_p=p, drop_seed=seed)
            dq = hip.attn_bwd(qkv, out, dout, lse, batch, L, H, 0.125, bias, drop_p=p, drop_seed=seed)
            with hip.option("attn_bwd", 2):
                dq2 = hip.attn_bwd(qkv, out, dout, lse, batch, L, H, 0.125, bias, drop_p=p, drop_seed=seed)
        res.append((out, lse, dq, dq2))
    for a, b2 in zip(res[0], res[1]):
        assert torch.equal(a, b2)


def test_embedding_dropout_and_ln_bwd_mask():
    hip = _hip()
    D, p, seed = 768, 0.1, 999
    ids = torch.randint(0, 300, (3, 40))
    word, pos, typ = rnd(300, D, seed=150), rnd(64, D, seed=151), rnd(2, D, seed=152)
    g, b = 1 + 0.1 * rnd(D, seed=153), 0.1 * rnd(D, seed=154)
    keep = _keep_mask(seed, 120 * D, p).view(120, D).double()
    y32, _ = hip.bert_embed(ids.cuda(), word.cuda(), pos.cuda(), typ.cuda(), g.cuda(), b.cuda(), 1e-12, torch.float32, drop_p=p, drop_seed=seed)
    e = (word[ids].double() + typ[0].double() + pos[:40].double()).view(120, D)
    e64 = e.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(e64, (D,), g.double(), b.double(), 1e-12) * keep / (1 - p)
    close(y32, ref, 1e-5, 1e-5, "embedding dropout")
    dy = rnd(120, D, seed=155)
    (ref * dy.double()).sum().backward()
    dx = torch.empty(120, D).cuda()
    hip.layernorm_bwd(dy.cuda(), e.float().cuda(), g.cuda(), 1e-12, dx, torch.zeros(D).cuda(), torch.zeros(D).cuda(), accumulate=False,
                      drop_p=p, drop_seed=seed)
    close(dx, e64.grad, 1e-4, 2e-4, "ln bwd through dropout")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_softmax_xent(dt):
    hip = _hip()
    M, V = 83, 30522
    logits = rnd(M, V, seed=160) * 2
    labels = torch.randint(0, V, (M,))
    labels[::3] = -100
    l64 = logits.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(l64, labels, reduction="none", ignore_index=-100)
    n = (labels != -100).sum()
    (ref.sum() / n).backward()
    inv_n = (1.0 / n.float()).reshape(1).cuda()
    loss_rows, dl = hip.softmax_xent(logits.cuda(), labels.cuda(), grad_dtype=dt, grad_scale=inv_n)
    close(loss_rows, ref.detach(), 1e-5, 1e-5, "xent rows")
    assert dl.shape == (M, 30528) and float(dl[:, V:].abs().sum()) == 0
    tol = (1e-5, 1e-8) if dt == torch.float32 else (1e-2, 1e-7)
    close(dl[:, :V], l64.grad, *tol, "xent grad")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K,tile", [(300, 3072, 768, "128"), (70000, 768, 768, "256"), (1000, 256, 768, "256")])
def test_gemm_gelu_bwd_epilogue(dt, M, N, K, tile):
    """alpro_gemm with ACT_GELU_BWD (dX *= gelu'(saved pre-activation)) on both tile kernels incl. partial edge tiles."""
    hip = _hip()
    dy = rnd(M, K, seed=300, scale=0.5).to(dt)
    w = rnd(N, K, seed=301, scale=0.05).to(dt)
    pre = rnd(M, N, seed=302).to(dt)
    with hip.option("gemm_tile", int(tile)):
        out = hip.gemm(dy.cuda(), w.cuda(), act=hip.ACT_GELU_BWD, pre_act=pre.cuda())
    p64 = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(p64).sum().backward()
    ref = (dy.double() @ w.double().T) * p64.grad
    rt, at = {torch.float32: (1e-4, 1e-4), torch.bfloat16: (1.5e-2, 1.5e-2), torch.float16: (3e-3, 3e-3)}[dt]
    close(out, ref, rt, at, "gelu-bwd epilogue")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K,tile", [(300, 3072, 768, "128"), (70000, 768, 768, "256"), (1000, 256, 768, "256")])
def test_gemm_gelu_save_grad_and_mul_saved_epilogues(dt, M, N, K, tile):
    """Round 3: the GELU Linear's forward keeps gelu'(pre-activation) (ACT_GELU_SAVE_GRAD writes it into C2 next to gelu(..)) and its dgrad
    multiplies by the saved factor (ACT_MUL_SAVED) -- both tile kernels incl. partial edge tiles, against fp64 autograd of erf-GELU."""
    hip = _hip()
    a = rnd(M, K, seed=320, scale=0.5).to(dt)
    w = rnd(N, K, seed=321, scale=0.08).to(dt)
    bias = rnd(N, seed=322, scale=0.3)
    rt, at = {torch.float32: (1e-4, 1e-4), torch.bfloat16: (1.5e-2, 1.5e-2), torch.float16: (3e-3, 3e-3)}[dt]
    saved = torch.empty(M, N, dtype=dt, device="cuda")
    with hip.option("gemm_tile", int(tile)):
        out = hip.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), act=hip.ACT_GELU_SAVE_GRAD, pre_act=saved)
    pre = (a.double() @ w.double().T + bias.double()).requires_grad_(True)
    y = torch.nn.functional.gelu(pre)
    y.sum().backward()
    close(out, y.detach(), rt, at, "gelu forward")
    close(saved, pre.grad, rt, at, "saved gelu'")
    dy = rnd(M, K, seed=323, scale=0.5).to(dt)
    with hip.option("gemm_tile", int(tile)):
        dx = hip.gemm(dy.cuda(), w.cuda(), act=hip.ACT_MUL_SAVED, pre_act=saved)
    close(dx, (dy.double() @ w.double().T) * saved.double().cpu(), rt, at, "mul-saved epilogue")
    with pytest.raises(RuntimeError, match="C2 buffer"):
        hip.gemm(dy.cuda(), w.cuda(), act=hip.ACT_MUL_SAVED)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gather_cast_colsum(dt):
    hip = _hip()
    B, T, N = 2, 4, 9
    dx = rnd(B, 1 + N * T, 768, seed=310)
    cs = torch.ones(768).cuda()
    rs = torch.rand(B * T).cuda()
    out = hip.gather_cast(dx.cuda(), dt, rows=B * T * (N + 1), map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N, row_scale=rs,
                          row_scale_group=N + 1, cls_scale=1.0 / T, colsum=cs)
    ref = hip.gather_cast(dx.cuda(), torch.float32, rows=B * T * (N + 1), map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N, row_scale=rs,
                          row_scale_group=N + 1, cls_scale=1.0 / T)
    close(out, ref.cpu().double(), *((1e-6, 1e-6) if dt == torch.float32 else (1e-2, 1e-2)), "gather_cast")
    close(cs, 1 + ref.cpu().double().sum(0), 1e-4, 1e-4, "gather_cast colsum")


def test_fused_qkv_wgrad_through_flat_gradient_buffer():
    """Once FlatAdamW owns the gradients, BertLayer.backward takes the query/key/value weight+bias gradients with ONE
    (3H, H) TN GEMM into a view of the flat buffer; the result must equal the three separate GEMMs."""
    _hip()
    import types
    from alpro_amd import config as rt
    from alpro_amd.modeling import train as tr
    from alpro_amd.modeling.xbert import BertLayer
    from alpro_amd.optim import FlatAdamW
    from tests.conftest import BERT_CFG
    cfg = types.SimpleNamespace(**dict(BERT_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, chunk_size_feed_forward=0))
    torch.manual_seed(3)
    layer = BertLayer(cfg, 0).cuda().train()
    B, L = 3, 40
    h32 = rnd(B * L, 768, seed=400).cuda()
    do32 = rnd(B * L, 768, seed=401).cuda()
    lins = (layer.attention.self.query, layer.attention.self.key, layer.attention.self.value)

    def run():
        with rt.use_compute_dtype(torch.bfloat16), torch.no_grad():
            _, _, sv = layer.forward_train(h32, h32.to(torch.bfloat16), None, B, L)
            layer.backward(sv, do32.clone(), None)
        return [l.weight.grad.clone() for l in lins] + [l.bias.grad.clone() for l in lins]

    sep = run()                                   # separate .grad tensors -> three GEMMs
    assert tr.fused_grad_view([l.weight for l in lins]) is None
    opt = FlatAdamW(layer.parameters(), lr=0.0)   # lr 0: the step only moves the gradients into the flat buffer
    opt.step()
    opt.zero_grad()
    assert tr.fused_grad_view([l.weight for l in lins]).shape == (3 * 768, 768)
    assert tr.fused_grad_view([l.bias for l in lins]).shape == (3 * 768,)
    fused = run()
    for a, b in zip(sep, fused):
        close(b, a.cpu().double(), 1e-5, 1e-4, "fused qkv wgrad")


def test_flat_adamw_behind_the_hvd_facade():
    """hvd.DistributedOptimizer(FlatAdamW) driven like run_pretrain_sparse.py:596-648 equals FlatAdamW.step() alone."""
    _hip()
    import sys
    import alpro_amd.compat
    sys.path.insert(0, alpro_amd.compat.PATH)
    from horovod import torch as hvd
    from alpro_amd.optim import FlatAdamW
    torch.manual_seed(1)
    a = [torch.nn.Parameter(torch.randn(9, 5).cuda()), torch.nn.Parameter(torch.randn(7).cuda())]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = FlatAdamW(a, lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=1.0)
    ob = hvd.DistributedOptimizer(FlatAdamW(b, lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=1.0))
    for step in range(3):
        gs = [torch.randn_like(p) for p in a]
        for ps in (a, b):
            for p, g in zip(ps, gs):
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
        oa.step()
        ob.synchronize()
        with ob.skip_synchronize():
            ob.step()
        for p, q_ in zip(a, b):
            close(q_, p.detach().cpu().double(), 1e-6, 1e-7, "facade step %d" % step)
        oa.zero_grad()
        ob.zero_grad()


@pytest.mark.parametrize("scenario", ["release", "decay"])
@pytest.mark.parametrize("drive", ["fused", "driver_order"])
def test_flat_adamw_vs_the_reference_optimizer_trajectory(scenario, drive):
    """a22 / N2 pinned to the reference: three steps of the REFERENCE's AdamW + get_lr_sched + torch clip_grad_norm_ on
    closed-form parameters and gradients (tests/golden/optimizer_adamw_3steps.npz, written by make_golden.case_optimizer from
    src/optimization/adamw.py / sched.py) against alpro_sumsq + alpro_adamw_step.  'fused': FlatAdamW(max_grad_norm=...) clips inside the
    step kernel.  'driver_order': the sequence of run_pretrain_sparse.py:596-646 through the hvd / amp facades -- backward, zero_none_grad
    (placeholder for the never-trained teacher), synchronize, lr of the step, torch's clip over amp.master_params (ONE flat view),
    the none-grad assertion, skip_synchronize / step / zero_grad."""
    _hip()
    import sys
    import numpy as np
    import alpro_amd.compat
    sys.path.insert(0, alpro_amd.compat.PATH)
    from horovod import torch as hvd
    from alpro_amd import amp, config as rt, optim
    from tests.conftest import GOLDEN
    from tests.golden.det_init import OPT_SCENARIOS
    from tests.test_host_cpu import _OptToy
    import os
    g = np.load(os.path.join(GOLDEN, "optimizer_adamw_3steps.npz"))
    hp = OPT_SCENARIOS[scenario]
    model = _OptToy("cuda")
    with rt.use_compute_dtype("fp32"):
        if drive == "fused":
            opt = optim.FlatAdamW(model.parameters(), lr=hp["lr"], betas=hp["betas"], weight_decay=hp["weight_decay"], max_grad_norm=hp["grad_norm"])
        else:
            opt = hvd.DistributedOptimizer(optim.FlatAdamW(model.parameters(), lr=hp["lr"], betas=hp["betas"], weight_decay=hp["weight_decay"]),
                                           named_parameters=model.named_parameters(), compression=hvd.Compression.none)
        for step in range(3):
            lr = float(g["%s/lr/%d" % (scenario, step)])
            loss = model.loss(step)
            if drive == "fused":
                loss.backward()
                opt.param_groups[0]["lr"] = lr
                opt.step()
                total = math.sqrt(float(opt.last_grad_norm))
                opt.zero_grad()
            else:
                with amp.scale_loss(loss, opt, delay_unscale=False) as scaled:
                    scaled.backward()
                    optim.zero_none_grad(model)
                    opt.synchronize()
                for pg in opt.param_groups:
                    pg["lr"] = lr
                views = list(amp.master_params(opt))
                assert len(views) == (len(model.ps) if step == 0 else 1)      # separate tensors before the first step built the flat buffers
                total = float(torch.nn.utils.clip_grad_norm_(views, hp["grad_norm"]))
                assert not [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
                with opt.skip_synchronize():
                    opt.step()
                    opt.zero_grad()
                assert optim.is_placeholder_grad(model.teacher.grad)
            assert total == pytest.approx(float(g["%s/grad_norm/%d" % (scenario, step)]), rel=2e-6)
            got = torch.cat([p.detach().reshape(-1) for p in model.ps]).cpu().numpy().astype(np.float64)
            np.testing.assert_allclose(got, g["%s/params/%d" % (scenario, step)], rtol=3e-6, atol=2e-8, err_msg="%s / %s step %d" % (scenario, drive, step))
        inner = getattr(opt, "_opt", opt)
        assert inner.n_params == sum(p.numel() for p in model.ps) and torch.equal(model.teacher.detach().cpu(), torch.ones(300, 7))
        where = {id(p): (o, p.numel()) for p, o in zip(inner.flat["live"], inner.flat["offs"])}     # flat offset of every trained parameter
        m = torch.cat([inner.flat["m"][where[id(p)][0]:sum(where[id(p)])] for p in model.ps]).cpu().numpy()
        v = torch.cat([inner.flat["v"][where[id(p)][0]:sum(where[id(p)])] for p in model.ps]).cpu().numpy()
        np.testing.assert_allclose(m, g[scenario + "/exp_avg"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(v, g[scenario + "/exp_avg_sq"], rtol=2e-5, atol=1e-10)


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-4), ("bf16", 4e-2)])
def test_merged_temporal_projection_equals_two_linears(mode, tol):
    """Block with merge_temporal_proj (one GEMM with We = Wfc Wp, product rule in backward) vs the two Linears of
    vit.py:157-162, train mode with a fixed drop-path pattern: same output, same input gradient, same parameter gradients."""
    _hip()
    from alpro_amd import config as rt
    from alpro_amd.modeling.timesformer.vit import Block
    torch.manual_seed(11)
    blk = Block(dim=768, num_heads=12, layer_num=0, mlp_ratio=4.0, qkv_bias=True, drop_path=0.3, attention_type='divided_space_time').cuda().train()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.3)
    B, T, N = 2, 4, 9
    S = 1 + N * T
    x = rnd(B, S, 768, seed=500).cuda()
    dout = rnd(B, S, 768, seed=501).cuda()
    masks = {B * N: (torch.rand(B * N, generator=torch.Generator().manual_seed(1)) > 0.3).float().cuda() / 0.7,
             B * T: (torch.rand(B * T, generator=torch.Generator().manual_seed(2)) > 0.3).float().cuda() / 0.7,
             B: torch.tensor([1 / 0.7, 0.0]).cuda()}
    blk._drop = lambda rows, device: masks[rows]
    res = {}
    for merged in (False, True):
        blk.merge_temporal_proj = merged
        for p in blk.parameters():
            p.grad = None
        with rt.use_compute_dtype(mode), torch.no_grad():
            out, sv = blk.forward_train(x.clone(), B, T, 3)
            dx, _ = blk.backward(sv, dout.clone())
        res[merged] = (out.clone(), dx.clone(), {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None})
    ref, got = res[False], res[True]

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))
    assert rel(got[0], ref[0]) < tol and rel(got[1], ref[1]) < tol
    assert set(got[2]) == set(ref[2])
    for n in ref[2]:
        assert rel(got[2][n], ref[2][n]) < 5 * tol, (n, rel(got[2][n], ref[2][n]))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_layernorm_bwd_emit_equals_layernorm_bwd_then_gather_cast(dt):
    """alpro_layernorm_bwd_emit (round 3): the operand rows it emits are exactly what alpro_gather_cast builds from the gradient stream
    the plain backward leaves behind -- for the three hand-overs of a divided space-time block (norm2 -> spatial projection in
    frame-token order with the CLS / T rule, norm1 -> temporal projection with the temporal_fc bias gradient, temporal norm -> the
    previous block's MLP incl. the CLS rows it never touches) and BERT's dropout hand-over, from 16-bit and from fp32 incoming
    gradients."""
    from alpro_amd import hip
    hip.load()
    B, T, N, D = 2, 4, 9, 768
    S = 1 + N * T
    x = (rnd(B, S, D, seed=700) * 2 + 0.3).cuda()
    g = (1 + 0.1 * rnd(D, seed=701)).cuda()
    dx0 = rnd(B, S, D, seed=702).cuda()
    sc_bt = ((torch.rand(B * T, generator=torch.Generator().manual_seed(3)) > 0.3).float() / 0.7).cuda()
    sc_bn = ((torch.rand(B * N, generator=torch.Generator().manual_seed(4)) > 0.3).float() / 0.7).cuda()
    sc_b = torch.tensor([1 / 0.7, 0.0]).cuda()

    def run(dy, rows, map_kw, emit, gather_kw, dy2=None):
        outs = []
        for use_emit in (False, True):
            dx = dx0.clone()
            dg, db, cs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
            if use_emit:
                e = dict(emit, dtype=dt)
                if "colsum_pre" in e:
                    e["colsum_pre"] = cs
                _, op = hip.layernorm_bwd(dy, x, g, 1e-6, dx, dg, db, rows=rows, dy2=dy2, emit=e, **map_kw)
            else:
                hip.layernorm_bwd(dy, x, g, 1e-6, dx, dg, db, rows=rows, dy2=dy2, **map_kw)
                gk = dict(gather_kw)
                if gk.pop("want_colsum_pre", False):
                    gk["colsum_pre"] = cs
                op = hip.gather_cast(dx, dt, **gk)
            outs.append((dx, op, dg, db, cs))
        (dx_a, op_a, dg_a, db_a, cs_a), (dx_b, op_b, dg_b, db_b, cs_b) = outs
        assert op_a.shape == op_b.shape and op_b.dtype == dt
        return dx_a, dx_b, op_a, op_b, cs_a, cs_b

    # norm2 (identity rows) -> frame-token operand of the spatial projection
    dy = rnd(B * S, D, seed=710).to(dt).cuda()
    dx_a, dx_b, op_a, op_b, _, _ = run(dy, B * S, {}, dict(mode=hip.EMIT_FRAME, rows=B * T * (N + 1), T=T, N=N, scale=sc_bt),
                                       dict(rows=B * T * (N + 1), map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N, row_scale=sc_bt, row_scale_group=N + 1, cls_scale=1.0 / T))
    assert torch.equal(dx_a, dx_b) and torch.equal(op_a, op_b)
    # norm1 (frame-token gather, CLS rows accumulated atomically) -> x[:, 1:] operand of the merged temporal projection + temporal_fc bias gradient
    dy = rnd(B * T * (N + 1), D, seed=711).to(dt).cuda()
    fm = dict(map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N)
    dx_a, dx_b, op_a, op_b, cs_a, cs_b = run(dy, B * T * (N + 1), fm, dict(mode=hip.EMIT_SKIP_CLS, rows=B * N * T, T=T, N=N, scale=sc_bn, group=T, colsum_pre=True),
                                             dict(rows=B * N * T, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T, row_scale=sc_bn, row_scale_group=T, want_colsum_pre=True))
    assert torch.equal(dx_a[:, 1:], dx_b[:, 1:]) and torch.equal(op_a, op_b)
    assert float((dx_a[:, 0] - dx_b[:, 0]).abs().max()) < 1e-4 * float(dx_a[:, 0].abs().max())          # T atomic adds per CLS row: order-dependent rounding only
    assert float((cs_a - cs_b).abs().max()) < 1e-4 * float(cs_a.abs().max())
    # temporal norm (x[:, 1:] rows) -> all rows of the previous block's MLP operand, CLS rows included
    dy = rnd(B * N * T, D, seed=712).to(dt).cuda()
    dx_a, dx_b, op_a, op_b, _, _ = run(dy, B * N * T, dict(map_mode=hip.MAP_SKIP_CLS, map_p0=N * T),
                                       dict(mode=hip.EMIT_ROWS, rows=B * S, T=T, N=N, scale=sc_b, group=S, extra_cls=B), dict(row_scale=sc_b, row_scale_group=S))
    assert torch.equal(dx_a, dx_b) and torch.equal(op_a, op_b)
    # BERT: fp32 incoming gradient + second stream, not accumulating; emitted through the dense-output dropout
    dy32, dy2 = rnd(B * S, D, seed=713).cuda(), rnd(B * S, D, seed=714).cuda()
    outs = []
    for use_emit in (False, True):
        dx = torch.empty(B * S, D, device="cuda")
        dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        if use_emit:
            _, op = hip.layernorm_bwd(dy32, x, g, 1e-12, dx, dg, db, dy2=dy2, accumulate=False, emit=dict(mode=hip.EMIT_ROWS, rows=B * S, dtype=dt, drop_p=0.1, drop_seed=77))
        else:
            hip.layernorm_bwd(dy32, x, g, 1e-12, dx, dg, db, dy2=dy2, accumulate=False)
            op = hip.gather_cast(dx, dt, drop_p=0.1, drop_seed=77)
        outs.append((dx, op))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert 0.05 < float((outs[1][1] == 0).float().mean()) < 0.15


@pytest.mark.parametrize("mode", ["fp32", "bf16", "fp16"])
def test_merged_projection_bank_equals_per_block_upkeep(mode):
    """Round 3: _MergedTProjBank keeps the merged temporal projections of all 12 blocks in batched launches (alpro_transpose_batch,
    alpro_gemm_batch, alpro_tproj_small; product rule per group of four blocks).
    (1) refresh: W_e and W_e^T are BIT-identical to the per-block code's, b1 = W_fc b_p to summation order (1 ulp: a different order than
        rocBLAS gemv -- which is why whole-encoder outputs agree only to rounding noise, amplified by 12 blocks, and are not compared bitwise);
    (2) after an in-place parameter update (what an optimizer does) every block is refreshed;
    (3) product rule on given dW_e / db1 against fp64 arithmetic on the operands the kernels see;
    (4) end to end: gradients of the banked and the per-block backward agree; a deep copy of the encoder gets its own bank."""
    import copy
    from alpro_amd import amp, config as rt
    from alpro_amd.modeling.timesformer import vit
    from tests.test_host_cpu import VENC
    hip = _hip()
    torch.manual_seed(11)
    enc = vit.TimeSformer(dict(VENC, num_frm=2, drop_path_rate=0.0), input_format="RGB").cuda().train()
    with torch.no_grad():
        for blk in enc.model.blocks:            # the reference zero-initialises temporal_fc of blocks > 0: give every block a live branch
            blk.temporal_fc.weight.normal_(0, 0.02)
            blk.temporal_fc.bias.normal_(0, 0.02)
            blk.temporal_attn.proj.bias.normal_(0, 0.02)
    bank = enc.model._tproj_bank
    D = 768

    def per_block(blk, dt):
        vit.Block.batch_merged_tproj = False
        blk._ops._store.pop("t_merged", None)
        return {k: v.clone() for k, v in blk._merged_tproj(dt).items()}

    try:
        with rt.use_compute_dtype(mode), torch.no_grad():
            dt = rt.compute_dtype()
            for rnd_ in range(2):
                for i in (0, 5, 11):
                    blk = enc.model.blocks[i]
                    ref = per_block(blk, dt)
                    vit.Block.batch_merged_tproj = True
                    got = blk._merged_tproj(dt)
                    assert blk._bank() is bank
                    assert torch.equal(got["w"], ref["w"]) and torch.equal(got["wT"], ref["wT"]), (rnd_, i)
                    assert float((got["b1"] - ref["b1"]).abs().max()) <= 1e-6 * float(ref["b1"].abs().max()), (rnd_, i)
                for p in enc.parameters():      # (2)
                    p.mul_(1.01)
            if mode != "fp32":                  # (3)
                ws = bank.workspace(dt, torch.device("cuda", 0))
                ws.copy_(rnd(*ws.shape, seed=900) * 0.01)
                for blk in enc.model.blocks:
                    for p in (blk.temporal_fc.weight, blk.temporal_attn.proj.weight, blk.temporal_attn.proj.bias):
                        p.grad = torch.full_like(p, 0.5)
                bank.product_rule(4, 8, dt)
                bank.product_rule(8, 12, dt)
                for i in (4, 7, 11):
                    blk = enc.model.blocks[i]
                    wf, wp, bp = (t.detach().double().cpu() for t in (blk.temporal_fc.weight, blk.temporal_attn.proj.weight, blk.temporal_attn.proj.bias))
                    dWe, db1 = ws[i, :D * D].view(D, D).double().cpu(), ws[i, D * D:].double().cpu()
                    qd = lambda t: t.to(dt).double()  # noqa: E731
                    ref_fc = 0.5 + qd(dWe.float()) @ qd(wp.float()).T + torch.outer(db1, bp)
                    ref_p = 0.5 + qd(wf.float()).T @ qd(dWe.float())
                    ref_bp = 0.5 + wf.T @ db1
                    for got, ref_, what in ((blk.temporal_fc.weight.grad, ref_fc, "dW_fc"), (blk.temporal_attn.proj.weight.grad, ref_p, "dW_p"),
                                            (blk.temporal_attn.proj.bias.grad, ref_bp, "db_p")):
                        err = float((got.double().cpu() - ref_).abs().max())
                        assert err <= 2e-5 * max(1.0, float(ref_.abs().max())), (what, i, err)
                assert float((enc.model.blocks[0].temporal_fc.weight.grad - 0.5).abs().max()) == 0.0      # untouched groups stay untouched
        # (4) end to end
        x = torch.randn(2, 3, 2, 224, 224, device="cuda")

        def run(model, banked):
            vit.Block.batch_merged_tproj = banked
            for p in model.parameters():
                p.grad = None
            with rt.use_compute_dtype(mode):
                out = model.forward_features(x)
                loss = out.float().square().mean()
                if mode == "fp16":
                    sc = amp.LossScaler(init_scale=256.0, dynamic=False, device="cuda")
                    with rt.loss_scaling(sc):
                        (loss * 256.0).backward()
                else:
                    loss.backward()
            return out.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

        ref_out, ref_g = run(enc, False)
        out, g = run(enc, True)
        tol = {"fp32": 2e-4, "fp16": 2e-2, "bf16": 1e-1}[mode]     # 1-ulp differences in b1, amplified through 12 blocks of 16-bit roundings
        assert float((out - ref_out).abs().max()) <= tol * float(ref_out.abs().max())
        assert set(g) == set(ref_g)
        for n in g:
            err = float((g[n] - ref_g[n]).norm() / ref_g[n].norm().clamp_min(1e-20))
            assert err <= tol, (n, err)
        twin = copy.deepcopy(enc)               # a deep copy must not talk to the original's bank
        out3, _ = run(twin, True)
        assert twin.model._tproj_bank is not bank and twin.model._tproj_bank.state is not None and twin.model.blocks[3]._bank() is twin.model._tproj_bank
        assert float((out3 - out).abs().max()) <= tol * float(out.abs().max())
    finally:
        vit.Block.batch_merged_tproj = True


def test_gemms_at_full_benchmark_size():
    """The three GEMM kernels at the default benchmark's sizes (B=64 x 8f: M = 100416 token rows) against torch.matmul on the
    same bf16 operands in fp32: many persistent rounds, a partial last M-tile, 27-tile XCD slices."""
    hip = _hip()
    dt = torch.bfloat16
    M = 100416
    g = torch.Generator(device="cuda").manual_seed(9)
    a = (torch.randn(M, 768, device="cuda", generator=g) * 0.5).to(dt)
    w = (torch.randn(2304, 768, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(2304, device="cuda", generator=g)
    out = hip.gemm(a, w, bias=bias)                                                     # qkv: bf16 out
    ref = a.float() @ w.float().t() + bias
    assert float((out.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    res = torch.randn(M, 768, device="cuda", generator=g)
    a2 = (torch.randn(M, 3072, device="cuda", generator=g) * 0.3).to(dt)
    w2 = (torch.randn(768, 3072, device="cuda", generator=g) * 0.02).to(dt)
    rs = (torch.rand(64, device="cuda", generator=g) > 0.1).float() / 0.9
    out2 = hip.gemm(a2, w2, out_dtype=torch.float32, residual=res, row_scale=rs, row_scale_group=M // 64)   # fc2 + residual, drop-path
    ref2 = res + rs.repeat_interleave(M // 64)[:, None] * (a2.float() @ w2.float().t())
    assert float((out2 - ref2).abs().max()) < 1e-3 * float(ref2.abs().max())
    dy = (torch.randn(M, 3072, device="cuda", generator=g) * 0.1).to(dt)
    gw = torch.zeros(3072, 768, device="cuda")
    gb = torch.zeros(3072, device="cuda")
    hip.gemm_tn_acc(dy, a, gw, colsum=gb)                                               # fc1 weight + bias gradient
    refw = dy.float().t() @ a.float()
    assert float((gw - refw).abs().max()) < 2e-3 * float(refw.abs().max())
    assert float((gb - dy.float().sum(0)).abs().max()) < 1e-3 * float(dy.float().sum(0).abs().max()) + 1e-2


def test_attention_properties_at_full_size():
    """Spatial and temporal attention at the benchmark's sizes (512 sequences x 197 tokens; 100352 rows in groups of 8) through
    a size-independent property: with V == 1 every softmax row must give exactly 1 (rows sum to one), and then dQ = dK = 0 and
    dV[k] = sum over queries of P[q, k] * dO[q], whose total over k equals the total of dO."""
    hip = _hip()
    dt = torch.bfloat16
    H = 12
    g = torch.Generator(device="cuda").manual_seed(21)
    for kind in ("spatial", "temporal"):
        rows = 512 * 197 if kind == "spatial" else 64 * 196 * 8
        qkv = (torch.randn(rows, 3 * H * 64, device="cuda", generator=g) * 0.5).to(dt)
        qkv[:, 2 * H * 64:] = 1.0
        do = torch.randn(rows, H * 64, device="cuda", generator=g).to(dt)
        if kind == "spatial":
            out, lse = hip.attn(qkv, 512, 197, H, 0.125, want_lse=True)
            dqkv = hip.attn_bwd(qkv, out, do, lse, 512, 197, H, 0.125)
        else:
            out, lse = hip.attn_temporal(qkv, 8, H, 0.125, want_lse=True)
            dqkv = hip.attn_temporal_bwd(qkv, out, do, lse, 8, H, 0.125)
        assert float((out.float() - 1).abs().max()) < 8e-3, kind
        assert float(dqkv[:, :2 * H * 64].float().abs().max()) < 2e-2, kind
        dv_tot, do_tot = float(dqkv[:, 2 * H * 64:].float().sum()), float(do.float().sum())
        assert abs(dv_tot - do_tot) < 2e-3 * float(do.float().abs().sum()), kind


@pytest.mark.parametrize("B,world,rank,temp", [(3, 1, 0, 0.07), (64, 1, 0, 0.07), (5, 2, 1, 0.05), (64, 8, 3, 0.9), (7, 3, 2, 0.0001)])
def test_vtc_loss_fwd_bwd(B, world, rank, temp):
    """alpro_vtc_loss (forward + backward through the autograd node the models use) against the reference's arithmetic
    (alpro_models.py:113-128) in fp64: loss, both similarity matrices, gradients w.r.t. the local rows, the gathered rows and
    the temperature; positives at columns rank*B + i; temp outside [0.001, 0.5] is clamped like :80-81."""
    _hip()
    import torch.nn.functional as F
    from alpro_amd.modeling.alpro_models import _VtcLoss
    E, G = 256, B * world
    nrm = lambda x: F.normalize(x, dim=-1)  # noqa: E731
    v, t = nrm(rnd(B, E, seed=900 + B)), nrm(rnd(B, E, seed=901 + B))
    gv, gt = nrm(rnd(G, E, seed=902 + B)), nrm(rnd(G, E, seed=903 + B))
    gv[rank * B:(rank + 1) * B], gt[rank * B:(rank + 1) * B] = v, t
    tp = torch.tensor(temp)
    leaves = [x.clone().cuda().requires_grad_(True) for x in (v, t, gv, gt, tp)]
    loss, s_v2t, s_t2v = _VtcLoss.apply(*leaves, rank * B)
    (loss * 1.7).backward()
    r = [x.double().requires_grad_(True) for x in (v, t, gv, gt, tp)]
    tc = r[4].clamp(0.001, 0.5)
    rv2t, rt2v = r[0] @ r[3].t() / tc, r[1] @ r[2].t() / tc
    tgt = torch.zeros(B, G, dtype=torch.float64)
    tgt[:, rank * B:(rank + 1) * B] = torch.eye(B, dtype=torch.float64)
    rl = (-(F.log_softmax(rv2t, 1) * tgt).sum(1).mean() - (F.log_softmax(rt2v, 1) * tgt).sum(1).mean()) / 2
    (rl * 1.7).backward()
    close(loss.reshape(1), rl.detach().reshape(1), 2e-5, 2e-5, "vtc loss")
    close(s_v2t, rv2t.detach(), 2e-5, 2e-4, "sim_v2t")
    close(s_t2v, rt2v.detach(), 2e-5, 2e-4, "sim_t2v")
    for got, ref, name in zip(leaves[:4], r[:4], ("dv", "dt", "dgv", "dgt")):
        close(got.grad, ref.grad, 2e-4, 2e-5 * float(ref.grad.abs().max().clamp_min(1.0)), name)
    if 0.001 < temp < 0.5:   # (a clamped temperature has zero gradient in the restatement, the in-place clamp_ of the model does not)
        close(leaves[4].grad.reshape(1), r[4].grad.reshape(1), 2e-4, 1e-4 * float(r[4].grad.abs()), "dtemp")


def test_wgrad_run_to_run_reproducibility():
    """Two ways of combining the token ranges of a weight gradient.  Workspace mode (alpro_gemm_tn_acc_ws: partial tiles stored, then
    added in a fixed order) is BIT-reproducible at every size, bias gradient included.  Atomic mode (alpro_gemm_tn_acc) adds the same
    summands in an order that can differ between launches: reproducible only up to fp32 re-association -- two runs of a weight
    gradient at the benchmark size agree to 2e-6 of the gradient's scale (observed ~3e-7), far below the bf16 operand rounding
    (4e-3) -- and bit-reproducible when the tokens are not split (tn_splits = 1)."""
    hip = _hip()
    dt = torch.bfloat16
    M = 100416
    g = torch.Generator(device="cuda").manual_seed(77)
    dy = (torch.randn(M, 768, device="cuda", generator=g) * 0.1).to(dt)
    x = (torch.randn(M, 768, device="cuda", generator=g) * 0.5).to(dt)

    def run(atomic, rows=M):
        gw, gb = torch.zeros(768, 768, device="cuda"), torch.zeros(768, device="cuda")
        hip.gemm_tn_acc(dy[:rows], x[:rows], gw, colsum=gb, atomic=atomic)
        return gw, gb
    assert hip.load().alpro_gemm_tn_workspace_bytes(M, 768, 768) > 0          # the tokens ARE split at this size
    ws = [run(False) for _ in range(3)]
    for gw, gb in ws[1:]:
        assert torch.equal(gw, ws[0][0]) and torch.equal(gb, ws[0][1])
    runs = [run(True)[0] for _ in range(3)] + [ws[0][0]]
    scale = float(runs[0].abs().max())
    worst = max(float((runs[0] - r).abs().max()) for r in runs[1:])
    assert worst <= 2e-6 * scale, (worst, scale)
    ref = dy[:20000].float().t() @ x[:20000].float()
    one = []
    with hip.option("tn_splits", 1):
        for _ in range(2):
            one.append(run(True, 20000)[0])
    assert torch.equal(one[0], one[1])
    assert float((one[0] - ref).abs().max()) < 2e-3 * float(ref.abs().max())


def test_transpose_batch_and_operand_refresh():
    """alpro_transpose_batch == the per-matrix alpro_transpose on ragged shapes, and the optimizer-side refresh of the registered W^T
    operands (modeling/train.py) hands the next backward exactly what the lazy per-Linear path would have produced."""
    hip = _hip()
    from alpro_amd.modeling import train as tr
    from alpro_amd.modeling.weights import OperandCache
    from alpro_amd.optim import FlatAdamW
    g = torch.Generator(device="cuda").manual_seed(5)
    for dt in (torch.bfloat16, torch.float32):
        srcs = [torch.randn(r, c, device="cuda", generator=g) for r, c in ((768, 768), (2304, 768), (100, 70), (1, 64), (3072, 768), (65, 129))]
        outs = [torch.full((s.shape[1], (s.shape[0] + 63) // 64 * 64), 7.0, device="cuda", dtype=dt) for s in srcs]
        table, n, tiles = hip.transpose_jobs(list(zip(srcs, outs)))
        hip.transpose_batch(table, n, tiles, dt)
        for s, o in zip(srcs, outs):
            assert torch.equal(o, hip.transpose(s, out_dtype=dt, pad_to=64))
    lins = [torch.nn.Linear(768, 768).cuda() for _ in range(3)] + [torch.nn.Linear(768, 3072).cuda()]
    params = [p for l in lins for p in l.parameters()]
    opt = FlatAdamW(params, lr=1e-2, allreduce=False)
    cache = OperandCache()

    def operands():
        return [tr.transposed_operand(cache, "qkv^T", tuple(l.weight for l in lins[:3]), torch.bfloat16), tr.transposed_operand(cache, "fc^T", lins[3].weight, torch.bfloat16)]
    for step in range(3):
        for p in params:
            p.grad = torch.randn_like(p) if p.grad is None else p.grad.copy_(torch.randn_like(p))
        before = [o.clone() for o in operands()]
        opt.step()
        got = operands()
        want = [hip.transpose(torch.cat([l.weight.detach() for l in lins[:3]], 0), out_dtype=torch.bfloat16, pad_to=64),
                hip.transpose(lins[3].weight.detach(), out_dtype=torch.bfloat16, pad_to=64)]
        for a, b, c in zip(got, want, before):
            assert torch.equal(a, b) and not torch.equal(a, c)
    assert len(tr._WT_REGISTRY) >= 2 and tr.refresh_transposed_operands() >= 2


def _determinism(hip, on):
    import contextlib

    @contextlib.contextmanager
    def cm():
        prev = hip.deterministic()
        hip.set_deterministic(on)
        try:
            yield
        finally:
            hip.set_deterministic(prev)
    return cm()


def test_reductions_run_to_run_reproducibility():
    """Round 4: the remaining fp32 atomics of the training step have a fixed-order form, and it is the default.
      * alpro_layernorm_bwd at the benchmark's ViT shape under the FRAME_TOKENS scatter: the clip's CLS row (one term per frame, parked in the
        workspace and added in frame order by cls_rows_reduce_kernel) and dgamma / dbeta / colsum_pre (per-workgroup partials in the reduction workspace + colsum_reduce_kernel);
      * alpro_gather_cast's colsum / colsum_pre, alpro_sumsq, the position form of alpro_scatter_add_rows, the sorted index form;
      * alpro_vtc_loss_fwd / _bwd (loss and d temp finished by one workgroup).
    Each runs three times on the same inputs: bitwise equal.  The atomic forms (set_deterministic(False), NULL workspace) stay available
    and agree with the fixed-order results to fp32 re-association."""
    hip = _hip()
    B, T, N, D = 16, 8, 196, 768
    S = 1 + N * T
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, S, D, device="cuda", generator=g) * 2 + 0.3
    gam = 1 + 0.1 * torch.randn(D, device="cuda", generator=g)
    rows = B * T * (N + 1)
    dy = (torch.randn(rows, D, device="cuda", generator=g) * 0.1).to(torch.float16)
    res = torch.randn(B, S, D, device="cuda", generator=g)
    drop_t = (torch.rand(B * N, device="cuda", generator=g) > 0.1).float() / 0.9

    def ln(emit):
        dx, dg, db, cp = res.clone(), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        kw = dict(rows=rows, map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N)
        if emit:   # norm1's backward as vit.py issues it: the same scatter + the SKIP_CLS emit with the temporal_fc bias column sums
            kw["emit"] = dict(mode=hip.EMIT_SKIP_CLS, rows=B * N * T, T=T, N=N, scale=drop_t, group=T, colsum_pre=cp)
            out = hip.layernorm_bwd(dy, x, gam, 1e-6, dx, dg, db, **kw)
            return dx, dg, db, cp, out[1]
        hip.layernorm_bwd(dy, x, gam, 1e-6, dx, dg, db, **kw)
        return dx, dg, db
    for emit in (False, True):
        runs = [ln(emit) for _ in range(3)]
        for r in runs[1:]:
            for a, b in zip(r, runs[0]):
                assert torch.equal(a, b), "layernorm_bwd (emit=%s) is not bit-reproducible" % emit
        with _determinism(hip, False):
            at = ln(emit)
        for a, b in zip(at, runs[0]):
            assert float((a.float() - b.float()).abs().max()) <= 2e-5 * max(float(b.float().abs().max()), 1.0)
    # the CLS rows really carry T terms each: against fp64 on a slice
    x64 = x[:2].double().cpu().requires_grad_(True)
    lnr = torch.nn.functional.layer_norm(x64, (D,), gam.double().cpu(), torch.zeros(D, dtype=torch.float64), 1e-6)
    xs = lnr[:, 1:].reshape(2, N, T, D).permute(0, 2, 1, 3)
    y = torch.cat([lnr[:, :1].unsqueeze(1).expand(2, T, 1, D), xs], 2).reshape(-1, D)
    (y * dy[:2 * T * (N + 1)].double().cpu()).sum().backward()
    close(ln(False)[0][:2, 0], res[:2, 0].double().cpu() + x64.grad[:, 0], 1e-4, 5e-4, "CLS-row gradient (frame terms through the workspace)")

    src = torch.randn(B * S, D, device="cuda", generator=g)
    rs = torch.rand(B, device="cuda", generator=g)

    def gc():
        cs, cp = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        out = hip.gather_cast(src, torch.float16, row_scale=rs, row_scale_group=S, colsum=cs, colsum_pre=cp)
        return out, cs, cp
    runs = [gc() for _ in range(3)]
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))
    close(runs[0][2], src.double().sum(0).cpu(), 1e-4, 2e-3, "gather_cast colsum_pre (workspace)")
    with _determinism(hip, False):
        at = gc()
    close(at[1], runs[0][1].double().cpu(), 1e-4, 2e-3, "gather_cast colsum atomic vs workspace")

    flat = torch.randn(50_000_003, device="cuda", generator=g)[4:]          # (16-byte aligned, length not a multiple of 4)
    outs = []
    for _ in range(3):
        o = torch.zeros(1, device="cuda")
        hip.sumsq(flat, o)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert abs(float(outs[0]) - float(flat.double().pow(2).sum())) <= 1e-5 * float(outs[0])
    with _determinism(hip, False):
        o = torch.zeros(1, device="cuda")
        hip.sumsq(flat, o)
    assert abs(float(o) - float(outs[0])) <= 1e-5 * float(outs[0])

    # embedding tables: 5120 token rows, heavy duplicates ([CLS] / [SEP] / [MASK] / pad), pad row skipped
    L, V = 40, 3000
    ids = torch.randint(4, V, (128 * L,), device="cuda", generator=g)
    ids[::L] = 1
    ids[7::L] = 2
    ids[torch.rand(128 * L, device="cuda", generator=g) < 0.15] = 3
    ids[torch.rand(128 * L, device="cuda", generator=g) < 0.2] = 0          # pad
    de = torch.randn(128 * L, D, device="cuda", generator=g)
    ref = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids.cpu(), (de * (ids != 0).unsqueeze(1)).double().cpu())

    def emb():
        gw, gp = torch.zeros(V, D, device="cuda"), torch.zeros(512, D, device="cuda")
        hip.scatter_add_rows(de, ids, gw, skip_idx=0)
        hip.scatter_add_rows(de, None, gp, idx_mod=L)
        return gw, gp
    runs = [emb() for _ in range(3)]
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1])
    close(runs[0][0], ref, 1e-5, 1e-4, "word-embedding scatter (sorted)")
    assert float(runs[0][0][0].abs().max()) == 0.0                             # the pad row's lookup gradient stays zero
    close(runs[0][1][:L], de.double().cpu().view(128, L, D).sum(0), 1e-5, 1e-4, "position scatter (owner wave)")
    with _determinism(hip, False):
        at = emb()
    close(at[0], ref, 1e-5, 1e-4, "word-embedding scatter (atomic, skip_idx)")
    assert float(at[0][0].abs().max()) == 0.0

    Bv, E = 64, 256
    v, t = [torch.nn.functional.normalize(torch.randn(Bv, E, device="cuda", generator=g), dim=-1) for _ in range(2)]
    temp = torch.tensor([0.07], device="cuda")

    def vtc():
        loss, s1, s2, lse = hip.vtc_loss_fwd(v, t, v, t, temp, 0)
        grads = hip.vtc_loss_bwd(v, t, v, t, temp, 0, s1, s2, lse, torch.ones(1, device="cuda"))
        return (loss,) + tuple(grads)
    runs = [vtc() for _ in range(3)]
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_gather_seq_forward_and_backward(dt):
    """alpro_gather_seq_fwd / _bwd (round 5): the fusion batch [text_pool[ti[s]] ; video_pool[vi[s]]] against torch.cat of index-selected pools
    (what alpro_models.py:278-281,325-330,360-363 build), its fp32 and operand-dtype copies bitwise; the backward against autograd of the same
    expression in fp64 (repeated and unused pool rows included), run twice -- bitwise equal (fixed summation order)."""
    hip = _hip()
    Pt, Pv, Lt, Lv, D, S = 6, 3, 5, 9, 768, 11
    g = torch.Generator().manual_seed(77)
    text, video = torch.randn(Pt, Lt, D, generator=g), torch.randn(Pv, Lv, D, generator=g)
    ti = torch.tensor([0, 0, 3, 5, 2, 2, 2, 1, 5, 0, 3])          # row 4 unused
    vi = torch.tensor([2, 2, 2, 0, 0, 2, 0, 2, 2, 0, 2])          # row 1 unused
    ref = torch.cat([text[ti], video[vi]], dim=1)
    out32, out_t = hip.gather_seq(text.cuda(), video.cuda(), ti.cuda(), vi.cuda(), dt)
    assert torch.equal(out32.cpu().view(S, Lt + Lv, D), ref)
    if dt == torch.float32:
        assert out_t is None
    else:
        assert torch.equal(out_t.cpu().view(S, Lt + Lv, D), ref.to(dt))
    d32 = torch.randn(S * (Lt + Lv), D, generator=g)
    d_t = None if dt == torch.float32 else (torch.randn(S * (Lt + Lv), D, generator=g) * 0.1).to(dt)
    t64, v64 = text.double().requires_grad_(True), video.double().requires_grad_(True)
    up = d32.double() + (0 if d_t is None else d_t.double())
    (torch.cat([t64[ti], v64[vi]], dim=1).reshape(-1, D) * up).sum().backward()
    a = hip.gather_seq_bwd(d32.cuda(), None if d_t is None else d_t.cuda(), ti.cuda(), vi.cuda(), Pt, Pv, Lt, Lv)
    b = hip.gather_seq_bwd(d32.cuda(), None if d_t is None else d_t.cuda(), ti.cuda(), vi.cuda(), Pt, Pv, Lt, Lv)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.allclose(a[0].cpu().double(), t64.grad, rtol=1e-6, atol=1e-6) and torch.allclose(a[1].cpu().double(), v64.grad, rtol=1e-6, atol=1e-6)
    assert float(a[0][4].abs().sum()) == 0.0 and float(a[1][1].abs().sum()) == 0.0      # unused pool rows get zeros, not garbage
