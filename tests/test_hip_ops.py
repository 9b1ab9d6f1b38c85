"""GPU: every C-ABI kernel against the oracle's arithmetic (torch fp64 on CPU) on seeded inputs.

16-bit storage dtypes: inputs are pre-rounded to the dtype so both sides see identical operands;
the remaining difference is fp32 accumulation order (+ one output rounding when the output is
16-bit), hence the tolerances below.  f32 is the exact mode: fp32 MFMA == fmaf chain.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

torch.backends.cuda.matmul.allow_tf32 = False


def _hip():
    from alpro_amd import hip
    hip.load()
    return hip


DTYPES = [torch.float32, torch.bfloat16, torch.float16]
OUT_TOL = {torch.float32: (2e-5, 2e-5), torch.bfloat16: (1e-2, 1e-2), torch.float16: (2e-3, 2e-3)}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(x, dt):
    return x.to(dt).to(torch.float64)


def close(got, ref, rtol, atol, what=""):
    got = got.detach().cpu().to(torch.float64)
    ref = ref.to(torch.float64)
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), "%s: %d/%d mismatches, max err %.3e (ref max %.3e)" % (what, int(bad.sum()), bad.numel(), float(err.max()), float(ref.abs().max()))


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2.0)))


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 200, 768), (128, 256, 3072), (77, 1002, 768), (1, 2, 768)])
def test_gemm_bias_act(dt, M, N, K):
    hip = _hip()
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    ref = q(a, dt) @ q(w, dt).T + b.double()
    for act, f in ((hip.ACT_NONE, lambda x: x), (hip.ACT_GELU, gelu), (hip.ACT_RELU, torch.relu)):
        out = hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), bias=b.cuda(), act=act, out_dtype=torch.float32)
        close(out, f(ref), 2e-5, 2e-4 if dt != torch.float32 else 2e-5, "gemm f32-out act=%d" % act)
    out = hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), bias=b.cuda(), act=hip.ACT_GELU)
    assert out.dtype == dt
    close(out, gelu(ref), *OUT_TOL[dt], "gemm T-out")


def test_gemm_transpose_detecting():
    """A = I (padded) with an asymmetric W must reproduce W^T exactly: catches row/col swaps in the C layout."""
    hip = _hip()
    K = 768
    a = torch.zeros(130, K)
    a[torch.arange(130), torch.arange(130)] = 1.0
    w = (torch.arange(200 * K, dtype=torch.float32).reshape(200, K) % 251) - 125.0
    out = hip.gemm(a.cuda(), w.cuda(), out_dtype=torch.float32)
    assert torch.equal(out.cpu(), w[:, :130].T.contiguous())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_residual_rowscale_alpha(dt):
    hip = _hip()
    M, N, K = 260, 768, 768
    a, w, b, r = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.05), rnd(N, seed=6), rnd(M, N, seed=7)
    rs = (torch.rand(M // 65 + 1) > 0.3).float() / 0.7
    ref = r.double() + rs.double().repeat_interleave(65)[:M, None] * (0.5 * (q(a, dt) @ q(w, dt).T) + b.double())
    out = hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), bias=b.cuda(), alpha=0.5, out_dtype=torch.float32, row_scale=rs.cuda(), row_scale_group=65,
                   residual=r.cuda())
    close(out, ref, 2e-5, 2e-4, "gemm residual")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_row_maps(dt):
    """SKIP_CLS / FRAME_TOKENS / PATCH_EMBED row maps against the reference's rearranges (vit.py:146-196,349-361)."""
    hip = _hip()
    B, T, N, D = 2, 4, 9, 768
    S = 1 + N * T
    w, bias = rnd(D, D, seed=8, scale=0.05), rnd(D, seed=9)
    x = rnd(B, S, D, seed=10)
    # SKIP_CLS: rows enumerate x[:, 1:]; out = x[:, 1:] + lin(a)
    a = rnd(B * N * T, D, seed=11)
    out = x.clone().cuda()
    hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), out=out.view(B * S, D), bias=bias.cuda(), out_dtype=torch.float32, residual=x.cuda().view(B * S, D),
             map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
    lin = (q(a, dt) @ q(w, dt).T + bias.double()).view(B, N * T, D)
    ref = x.double().clone()
    ref[:, 1:] += lin
    close(out, ref, 2e-5, 2e-4, "skip_cls")
    # FRAME_TOKENS scatter: rows (b t)(1+n); CLS rows -> side, patches -> x[b, 1 + n*T + t] + lin
    a = rnd(B * T * (N + 1), D, seed=12)
    out = torch.zeros(B, S, D).cuda()
    side = torch.zeros(B * T, D).cuda()
    hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), out=out.view(B * S, D), bias=bias.cuda(), out_dtype=torch.float32, residual=x.cuda().view(B * S, D),
             map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N, side=side)
    lin = (q(a, dt) @ q(w, dt).T + bias.double()).view(B, T, N + 1, D)
    ref = torch.zeros(B, S, D, dtype=torch.float64)
    ref[:, 1:] = x.double()[:, 1:] + lin[:, :, 1:].permute(0, 2, 1, 3).reshape(B, N * T, D)
    close(out, ref, 2e-5, 2e-4, "frame_tokens patches")
    close(side, lin[:, :, 0].reshape(B * T, D), 2e-5, 2e-4, "frame_tokens cls side")
    # PATCH_EMBED: rows (b t) n -> x[b, 1 + n*T + t] = lin + table[n*T + t]
    a = rnd(B * T * N, D, seed=13)
    table = rnd(N * T, D, seed=14)
    out = torch.zeros(B, S, D).cuda()
    hip.gemm(a.to(dt).cuda(), w.to(dt).cuda(), out=out.view(B * S, D), out_dtype=torch.float32, residual=table.cuda(), map_mode=hip.MAP_PATCH_EMBED,
             map_p0=T, map_p1=N)
    lin = (q(a, dt) @ q(w, dt).T).view(B, T, N, D).permute(0, 2, 1, 3).reshape(B, N * T, D)
    ref = torch.zeros(B, S, D, dtype=torch.float64)
    ref[:, 1:] = lin + table.double()
    close(out, ref, 2e-5, 2e-4, "patch_embed")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_tail_split_is_bitwise_neutral(dt):
    """Round 3: the last, partly filled round of the persistent 256 x 256 GEMM is cut into half tiles (128 x 256, two workgroups per
    tile) when at most half of the workgroups would be busy.  Every output element is still one fp32 accumulation in the same order,
    so results with and without the split must be IDENTICAL -- 16-bit output (c16 epilogue), fp32 residual epilogue with a row scale,
    GELU with the saved derivative, and a ragged M (partial half tile through the predicated epilogue)."""
    hip = _hip()
    for (M, N, K, kind) in [(50176, 768, 768, "c16"), (50208, 768, 3072, "res"), (50208, 3072, 768, "gelu"), (50000, 768, 768, "c16"), (32100, 768, 768, "res")]:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        assert 0 < tiles % 256 <= 128, (M, N, tiles)          # these shapes do take the split path
        a = rnd(M, K, seed=60, scale=0.5).to(dt).cuda()
        w = rnd(N, K, seed=61, scale=0.05).to(dt).cuda()
        bias = rnd(N, seed=62, scale=0.3).cuda()
        outs = []
        for tail in (0, 1):
            with hip.option("gemm_tail", tail), hip.option("gemm_kind", 0):   # (the round-3 kernel: on the 8-phase kernel gemm_tail is the round-6 hand-over below)
                if kind == "c16":
                    outs.append((hip.gemm(a, w, bias=bias),))
                elif kind == "res":
                    res = rnd(M, N, seed=63).cuda()
                    rs = (torch.rand(M // 16 + 1, generator=torch.Generator().manual_seed(5)) + 0.5).cuda()
                    outs.append((hip.gemm(a, w, bias=bias, out_dtype=torch.float32, residual=res, row_scale=rs, row_scale_group=16),))
                else:
                    saved = torch.empty(M, N, dtype=dt, device="cuda")
                    outs.append((hip.gemm(a, w, bias=bias, act=hip.ACT_GELU_SAVE_GRAD, pre_act=saved), saved))
        for x0, x1 in zip(*outs):
            assert torch.equal(x0, x1), (M, N, K, kind)
        ref = a[-300:].double().cpu() @ w.double().cpu().T + bias.double().cpu()
        if kind == "c16":
            close(outs[1][0][-300:], ref, *OUT_TOL[dt], "tail rows vs fp64")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_tail_handover_to_the_small_tile_kernel(dt):
    """Round 6 (option gemm_tail on the 8-phase kernel): when the last round of 256 x 256 tiles would keep at most 40 % of the workgroups busy and
    K >= 2048 (at K = 768 the small kernel is no faster than the round it replaces: measured, csrc/gemm.hip), the row panels that make it up are
    computed by the 128 x 128 kernel in a second launch.  Same arithmetic per element, another summation order:
    with and without the hand-over the results agree to the output dtype's resolution (fp32 outputs: to 1e-5 of the row), the handed-over rows
    match fp64, two runs are bitwise equal, and every epilogue the model sends down this path survives the cut -- 16-bit output, fp32 residual +
    row scale (groups cut by the split point), inference GELU, drop-path row scale, a ragged last panel.  Shapes: the B = 32 long-K N = 768 shapes
    (588 / 591 tiles: fc2 and the dgrads of qkv / fc1), one whose remainder is too large and one whose K is too short to be handed over (those
    must not change at all)."""
    hip = _hip()
    cases = [(50176, 768, 3072, "c16", True), (50432, 768, 2304, "scale", True), (50208, 768, 3072, "res", True), (50176, 768, 2048, "gelu", True), (100352, 768, 3072, "c16", False),
             (50176, 768, 768, "c16", False)]
    for (M, N, K, kind, expect_split) in cases:
        tiles = ((M + 255) // 256) * (N // 256)
        R = tiles % 256
        assert (0 < R and 5 * R <= 2 * 256 and K >= 2048) == expect_split, (M, N, tiles, R)
        a = rnd(M, K, seed=70, scale=0.5).to(dt).cuda()
        w = rnd(N, K, seed=71, scale=0.05).to(dt).cuda()
        bias = rnd(N, seed=72, scale=0.3).cuda()
        kw = dict(bias=bias)
        if kind == "res":
            kw.update(out_dtype=torch.float32, residual=rnd(M, N, seed=73).cuda(), row_scale=(torch.rand(M // 1569 + 1, generator=torch.Generator().manual_seed(5)) + 0.5).cuda(),
                      row_scale_group=1569)
        elif kind == "scale":
            kw.update(row_scale=(torch.rand(M // 197 + 1, generator=torch.Generator().manual_seed(6)) + 0.5).cuda(), row_scale_group=197)
        elif kind == "gelu":
            kw.update(act=hip.ACT_GELU)
        outs = []
        for tail in (0, 1, 1):
            with hip.option("gemm_tail", tail):
                outs.append(hip.gemm(a, w, **kw))
        assert torch.equal(outs[1], outs[2]), (M, N, K, kind)
        if not expect_split:
            assert torch.equal(outs[0], outs[1]), (M, N, K, kind)
            continue
        d = (outs[0].float() - outs[1].float()).abs().max().item()
        lim = 1e-5 * max(1.0, outs[0].float().abs().max().item()) * (K / 768) if kind == "res" else {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dt] * max(1.0, outs[0].float().abs().max().item())
        assert d <= lim, (M, N, K, kind, d, lim)
        first_tail = (((M + 255) // 256) - (R + N // 256 - 1) // (N // 256)) * 256
        assert not torch.equal(outs[0][first_tail:], outs[1][first_tail:]) or d == 0.0        # (the hand-over did happen: the tail rows come from another kernel)
        assert torch.equal(outs[0][:first_tail], outs[1][:first_tail]), (M, N, K, kind)          # ... and nothing in front of it moved
        ref = a[-300:].double().cpu() @ w.double().cpu().T + bias.double().cpu()
        if kind == "c16":
            close(outs[1][-300:], ref, *OUT_TOL[dt], "handed-over rows vs fp64")


def test_gemm_rejects_bad_k():
    hip = _hip()
    with pytest.raises(RuntimeError, match="K="):
        hip.gemm(torch.zeros(4, 100).cuda(), torch.zeros(4, 100).cuda())


# ------------------------------------------------------------------------------------------------ LayerNorm & friends
@pytest.mark.parametrize("dt", DTYPES)
def test_layernorm_maps(dt):
    hip = _hip()
    B, T, N, D = 2, 4, 9, 768
    S = 1 + N * T
    x = rnd(B, S, D, seed=20) * 3 + 0.5
    g, b = 1 + 0.1 * rnd(D, seed=21), 0.1 * rnd(D, seed=22)
    ln = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6)
    y, y32, mean, rstd = hip.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, dt, out32=True, stats=True)
    close(y32, ln.view(-1, D), 1e-5, 1e-5, "ln identity f32")
    close(y, ln.view(-1, D), *OUT_TOL[dt], "ln identity T")
    close(mean, x.double().mean(-1).flatten(), 1e-5, 1e-5)
    close(rstd, 1 / torch.sqrt(x.double().var(-1, unbiased=False) + 1e-6).flatten(), 1e-5, 1e-5)
    y = hip.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, torch.float32, rows=B * N * T, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
    close(y, ln[:, 1:].reshape(-1, D), 1e-5, 1e-5, "ln skip_cls")
    y = hip.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, torch.float32, rows=B * T * (N + 1), map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N)
    xs = ln[:, 1:].reshape(B, N, T, D).permute(0, 2, 1, 3)
    ref = torch.cat([ln[:, :1].unsqueeze(1).expand(B, T, 1, D), xs], 2).reshape(-1, D)
    close(y, ref, 1e-5, 1e-5, "ln frame_tokens gather")


@pytest.mark.parametrize("dt", DTYPES)
def test_add_layernorm_modes(dt):
    """alpro_add_layernorm_fwd (round 3): the residual add of a branch fused into the LayerNorm that follows, in the four row
    arrangements of the path, against the reference's own tensor algebra (vit.py:157-200: rearranges, cat, CLS frame mean) in fp64."""
    hip = _hip()
    B, T, N, D = 2, 4, 9, 768
    S = 1 + N * T
    x = rnd(B, S, D, seed=40) * 2 + 0.3
    g, b = 1 + 0.1 * rnd(D, seed=41), 0.1 * rnd(D, seed=42)
    bias = 0.2 * rnd(D, seed=43)
    ln = lambda v: torch.nn.functional.layer_norm(v, (D,), g.double(), b.double(), 1e-6)  # noqa: E731
    xd = x.double()
    tol32 = (2e-5, 2e-5)
    # identity (BERT: s = h + dense(ctx); LN(s)) with the fp32 copy of the normalised rows
    d = rnd(B * S, D, seed=44).to(dt)
    y, y32, xo = hip.add_layernorm(x.cuda(), d.cuda(), g.cuda(), b.cuda(), 1e-6, out32=True)
    ref_x = xd.view(-1, D) + d.double()
    close(xo.view(-1, D), ref_x, *tol32, "identity x'")
    close(y32, ln(ref_x), *tol32, "identity y32")
    close(y, ln(ref_x), *OUT_TOL[dt], "identity y")
    # PRE_SPATIAL: xt[:, 1:] = x[:, 1:] + (delta + bias) in '(b n t)' order; CLS untouched; LN over the frame-token gather (vit.py:162-180)
    d = rnd(B * N * T, D, seed=45).to(dt)
    y, xo = hip.add_layernorm(x.cuda(), d.cuda(), g.cuda(), b.cuda(), 1e-6, mode=hip.ADD_PRE_SPATIAL, delta_bias=bias.cuda(), T=T, N=N)
    xt = xd.clone()
    xt[:, 1:] += d.double().view(B, N * T, D) + bias.double()
    close(xo, xt, *tol32, "pre_spatial x'")
    lx = ln(xt)
    xs = lx[:, 1:].reshape(B, N, T, D).permute(0, 2, 1, 3)
    ref = torch.cat([lx[:, :1].unsqueeze(1).expand(B, T, 1, D), xs], 2).reshape(-1, D)
    close(y, ref, *OUT_TOL[dt], "pre_spatial y (frame-token order)")
    # ... in place, and without keeping x' (the prompter's last block)
    xi = x.cuda().clone()
    y2, xo2 = hip.add_layernorm(xi, d.cuda(), g.cuda(), b.cuda(), 1e-6, mode=hip.ADD_PRE_SPATIAL, x_out=xi, delta_bias=bias.cuda(), T=T, N=N)
    assert xo2 is xi and torch.equal(xi, xo) and torch.equal(y2, y)
    y3, none = hip.add_layernorm(x.cuda(), d.cuda(), g.cuda(), b.cuda(), 1e-6, mode=hip.ADD_PRE_SPATIAL, want_x=False, delta_bias=bias.cuda(), T=T, N=N)
    assert none is None and torch.equal(y3, y)
    # PRE_MLP: delta in frame-token order '(b t) (1 + n)'; patches scattered back, CLS gets the frame mean (vit.py:184-196), LN2 over all tokens
    d = rnd(B * T * (N + 1), D, seed=46).to(dt)
    y, xo = hip.add_layernorm(x.cuda(), d.cuda(), g.cuda(), b.cuda(), 1e-6, mode=hip.ADD_PRE_MLP, T=T, N=N)
    dd = d.double().view(B, T, N + 1, D)
    x2 = xd.clone()
    x2[:, 0] += dd[:, :, 0].mean(1)
    x2[:, 1:] += dd[:, :, 1:].permute(0, 2, 1, 3).reshape(B, N * T, D)
    close(xo, x2, *tol32, "pre_mlp x'")
    close(y, ln(x2).view(-1, D), *OUT_TOL[dt], "pre_mlp y")
    # PRE_TEMPORAL: the MLP delta of the previous block folded into the next block's temporal LayerNorm over x[:, 1:]
    d = rnd(B * S, D, seed=47).to(dt)
    y, xo = hip.add_layernorm(x.cuda(), d.cuda(), g.cuda(), b.cuda(), 1e-6, mode=hip.ADD_PRE_TEMPORAL, T=T, N=N)
    x3 = xd + d.double().view(B, S, D)
    close(xo, x3, *tol32, "pre_temporal x'")
    close(y, ln(x3)[:, 1:].reshape(-1, D), *OUT_TOL[dt], "pre_temporal y (x[:, 1:] order)")
    with pytest.raises(RuntimeError, match="whole number of clips"):
        hip.add_layernorm(x.cuda()[:, :-1].contiguous(), d.cuda(), g.cuda(), b.cuda(), 1e-6, mode=hip.ADD_PRE_MLP, T=T, N=N)
    with pytest.raises(RuntimeError, match="delta has"):
        hip.add_layernorm(x.cuda(), d.cuda()[:-1], g.cuda(), b.cuda(), 1e-6, mode=hip.ADD_PRE_TEMPORAL, T=T, N=N)


def test_small_kernels():
    hip = _hip()
    B, T, N, D = 3, 4, 6, 768
    S = 1 + N * T
    x = rnd(B, S, D, seed=30)
    side = rnd(B * T, D, seed=31)
    out = torch.zeros(B, S, D).cuda()
    hip.cls_mean_residual(x.cuda(), side.cuda(), out, B, T)
    close(out[:, 0], x[:, 0].double() + side.double().view(B, T, D).mean(1), 1e-6, 1e-6, "cls mean")
    g, b = 1 + 0.1 * rnd(D, seed=32), 0.1 * rnd(D, seed=33)
    o32, ot = hip.vit_final_pool(x.cuda(), g.cuda(), b.cuda(), 1e-6, B, T, N, torch.bfloat16)
    ln = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6)
    ref = torch.cat([ln[:, :1], ln[:, 1:].reshape(B, N, T, D).mean(2)], 1)
    close(o32, ref, 1e-5, 1e-5, "final pool")
    close(ot, ref, 1e-2, 1e-2, "final pool bf16")
    img = rnd(2, 3, 32, 48, seed=34)
    p = hip.patchify(img.cuda(), torch.float32)
    ref = torch.nn.functional.unfold(img, 16, stride=16).transpose(1, 2).reshape(-1, 768)  # (c, i, j) fastest = conv weight order
    assert torch.equal(p.cpu(), ref)
    pb = hip.patchify(img.cuda(), torch.bfloat16)
    assert torch.equal(pb.cpu(), ref.to(torch.bfloat16))
    v = rnd(1000003, seed=35)
    assert torch.equal(hip.cast(v.cuda(), torch.bfloat16).cpu(), v.to(torch.bfloat16))
    assert torch.equal(hip.cast(v.cuda(), torch.float16).cpu(), v.to(torch.float16))
    ids = torch.randint(0, 500, (3, 40))
    word, pos, typ = rnd(500, D, seed=36), rnd(64, D, seed=37), rnd(2, D, seed=38)
    y32, yt = hip.bert_embed(ids.cuda(), word.cuda(), pos.cuda(), typ.cuda(), g.cuda(), b.cuda(), 1e-12, torch.bfloat16)
    e = word[ids].double() + typ[0].double() + pos[:40].double()
    ref = torch.nn.functional.layer_norm(e, (D,), g.double(), b.double(), 1e-12).view(-1, D)
    close(y32, ref, 1e-5, 1e-5, "bert embed")
    close(yt, ref, 1e-2, 1e-2, "bert embed bf16")


# ------------------------------------------------------------------------------------------------ attention
def ref_attention(qkv, batch, L, H, scale, bias=None, group=None):
    """qkv (batch*L, 3*H*64) float64 -> (batch*L, H*64); same arithmetic as vit.py:84-96 / xbert.py:299-341."""
    t = qkv.view(batch, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (t[0] @ t[1].transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias[:, None, None, :].double()
    if group is not None:
        idx = torch.arange(L) // group
        s = s.masked_fill(idx[:, None] != idx[None, :], float("-inf"))
    p = s.softmax(-1)
    return (p @ t[2]).transpose(1, 2).reshape(batch * L, H * 64), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("batch,L,masked", [(3, 40, True), (2, 197, False), (2, 237, True), (1, 100, False), (2, 256, False)])
def test_attn_full(dt, batch, L, masked):
    hip = _hip()
    H = 12
    qkv = rnd(batch * L, 3 * H * 64, seed=40 + L).to(dt)
    bias = None
    if masked:
        m = torch.ones(batch, L)
        for b in range(batch):
            m[b, L - 3 - 5 * b:] = 0
        bias = (1.0 - m) * -10000.0
    out, lse = hip.attn(qkv.cuda(), batch, L, H, 0.125, None if bias is None else bias.cuda(), want_lse=True)
    ref, ref_lse = ref_attention(qkv.double(), batch, L, H, 0.125, bias)
    tol = {torch.float32: (2e-5, 2e-5), torch.bfloat16: (2e-2, 2e-2), torch.float16: (3e-3, 3e-3)}[dt]  # P and O are rounded to dt
    close(out, ref, *tol, "attn out")
    close(lse, ref_lse, 1e-5, 1e-4, "attn lse")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,groups", [(8, 11), (4, 17), (2, 5), (16, 3), (8, 64)])
def test_attn_temporal(dt, T, groups):
    hip = _hip()
    H = 12
    rows = groups * T
    qkv = rnd(rows, 3 * H * 64, seed=60 + T).to(dt)
    out = hip.attn_temporal(qkv.cuda(), T, H, 0.125)
    ref, _ = ref_attention(qkv.double(), groups, T, H, 0.125)
    tol = {torch.float32: (2e-5, 2e-5), torch.bfloat16: (2e-2, 2e-2), torch.float16: (3e-3, 3e-3)}[dt]
    close(out, ref, *tol, "temporal attn")


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,H,T,K,with_bias", [(256 * 3, 12, 8, 768, True), (256 * 2 + 96, 12, 8, 768, True), (32, 2, 8, 128, False), (1568 * 2, 12, 8, 768, True),
                                               (512, 3, 2, 256, True), (320, 12, 4, 768, True), (256, 1, 16, 384, True), (64, 12, 1, 768, True)])
def test_gemm_qkv_temporal_attention_fused(dt, M, H, T, K, with_bias):
    """alpro_gemm_qkv_tattn (round 6; vit.py:84-98 on the temporal view of vit.py:152-156): the qkv Linear and the T-frame attention in one launch,
    q / k / v consumed out of the accumulators.  Against fp64 arithmetic on the operands as stored (q, k, v rounded to the storage dtype like the
    two-launch path stores them), and against that two-launch path itself (alpro_gemm + alpro_attn_temporal_fwd) -- same roundings, so the two
    agree to a few ulps of the output.  Ragged row panels (M % 256 != 0), every frame count the path allows, one and many heads, no bias."""
    hip = _hip()
    a = rnd(M, K, seed=700 + M).to(dt)
    w = rnd(3 * H * 64, K, seed=701 + H, scale=2.0 / K ** 0.5).to(dt)
    b = rnd(3 * H * 64, seed=702) if with_bias else None
    scale = 0.125
    out = hip.gemm_qkv_tattn(a.cuda(), w.cuda(), None if b is None else b.cuda(), T, H, scale)
    qkv = a.double() @ w.double().T + (0 if b is None else b.double())
    qkv = qkv.to(dt).double()                                   # the storage rounding of q, k, v
    ref, _ = ref_attention(qkv, M // T, T, H, scale)
    tol = {torch.bfloat16: (2e-2, 2e-2), torch.float16: (3e-3, 3e-3)}[dt]
    close(out, ref, *tol, "fused qkv + temporal attention vs fp64")
    two = hip.attn_temporal(hip.gemm(a.cuda(), w.cuda(), bias=None if b is None else b.cuda()), T, H, scale)
    d = (out.float() - two.float()).abs().max().item()
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dt] * max(1.0, two.float().abs().max().item())
    assert d <= 4 * ulp, "fused vs alpro_gemm + alpro_attn_temporal_fwd: %.3e (ulp %.1e)" % (d, ulp)
    again = hip.gemm_qkv_tattn(a.cuda(), w.cuda(), None if b is None else b.cuda(), T, H, scale)
    assert torch.equal(out, again)
    # the training form: the same output, plus q | k | v as the qkv Linear stores them and the log-sum-exp rows the temporal backward reads
    out3, qkv3, lse3 = hip.gemm_qkv_tattn(a.cuda(), w.cuda(), None if b is None else b.cuda(), T, H, scale, want_qkv=True)
    assert torch.equal(out3, out)
    g2 = hip.gemm(a.cuda(), w.cuda(), bias=None if b is None else b.cuda())
    dq = (qkv3.float() - g2.float()).abs().max().item()
    assert dq <= {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dt] * max(1.0, g2.float().abs().max().item()), "q | k | v vs alpro_gemm: %.3e" % dq
    _, lse2 = hip.attn_temporal(qkv3, T, H, scale, want_lse=True)
    assert lse3.shape == lse2.shape
    close(lse3, lse2.double().cpu(), 2e-5, 2e-5, "log-sum-exp rows vs alpro_attn_temporal_fwd on the same q | k | v")


@pytest.mark.parametrize("M,N,K,case", [(64, 768, 3072, "res_scale"), (3, 2304, 768, "ln"), (512, 768, 768, "scale"), (64, 3072, 768, "ln_gelu"), (130, 256, 768, "plain"), (1, 768, 768, "ln")])
def test_gemm_rows_f32(M, N, K, case):
    """alpro_gemm_rows_f32 (round 4: the fp32 Linears of the precise CLS-row chain, a few rows against a full fp32 weight, optional fused
    LayerNorm, exact erf GELU, row scale, residual) against fp64, with a row-strided A view and run twice (fixed summation order: bitwise equal)."""
    hip = _hip()
    wide = rnd(M, K + 64, seed=800 + M)
    a, w, b = wide[:, :K], rnd(N, K, seed=801, scale=0.05), rnd(N, seed=802)
    g, be = 1.0 + 0.1 * rnd(K, seed=803), 0.1 * rnd(K, seed=804)
    x = a.double()
    kw = {}
    if case.startswith("ln"):
        kw["ln"] = (g.cuda(), be.cuda(), 1e-6)
        x = torch.nn.functional.layer_norm(x, (K,), g.double(), be.double(), 1e-6)
    ref = x @ w.double().T + b.double()
    if case == "ln_gelu":
        kw["act"] = hip.ACT_GELU
        ref = gelu(ref)
    if case in ("res_scale", "scale"):
        rs = (torch.arange(M) % 3).float() * 0.75
        kw["row_scale"] = rs.cuda()
        ref = ref * rs.double()[:, None]
    if case == "res_scale":
        res = rnd(M, N, seed=805)
        kw["residual"] = res.cuda()
        ref = ref + res.double()
    A = wide.cuda()[:, :K]
    out = hip.gemm_rows(A, w.cuda(), bias=b.cuda(), **kw)
    close(out, ref, 2e-5, 2e-5 * math.sqrt(K / 768), "gemm_rows " + case)
    assert torch.equal(out, hip.gemm_rows(A, w.cuda(), bias=b.cuda(), **kw))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,case", [(256 * 80, 512, 768, "plain"), (256 * 54, 768, 256, "plain"), (256 * 130, 768, 768, "bias_rowscale"), (256 * 40, 1024, 3072, "f32res"),
                                        (256 * 80, 512, 768, "gelu_save"), (256 * 80, 512, 768, "mul_saved"), (256 * 80, 512, 768, "gelu"), (256 * 44, 1024, 768, "dropout"),
                                        (256 * 80, 512, 768, "strided"), (256 * 80 + 64, 512, 768, "plain"), (256 * 80 + 64, 512, 768, "gelu_save"), (256 * 392 + 64, 768, 3072, "mul_saved"), (64 * 1569, 768, 3072, "f32res_rowscale"), (16 * 1569 * 4, 768, 256, "f32res_rowscale"),
                                        (256 * 80 + 16, 512, 768, "bias_rowscale"), (256 * 80 + 240, 512, 768, "mul_saved"), (256 * 80 + 144, 1024, 256, "f32res"), (256 * 80 + 8, 512, 768, "plain")])
def test_gemm_8phase_kernel(dt, M, N, K, case):
    """gemm_nt256q_kernel (round 4: 8-phase two-group schedule on 16x16x32 MFMA fragments; option gemm_kind = 1) on full-tile shapes with >= 160
    tiles -- one and several tiles per workgroup, K = 4 / 12 / 48 K-tiles, every epilogue the identity-map shapes of the model use -- against
    fp64 of the same rounded operands, and against the round-3 kernel (gemm_kind = 0) on the same inputs.  Ragged M (the ViT's B * 1569 rows:
    M % 256 = 64): a remainder that is a multiple of 16 rows (16 / 64 / 144 / 240 here) is one more tile row of the SAME launch (invalid copy
    pieces re-read valid rows, fragment rows beyond M are not stored -- the rows behind M must stay untouched: checked through a guard band);
    any other remainder (+8) is split by the launcher: whole tiles on the 8-phase kernel, the rest on the 128 x 128 kernel (which indexes
    the row scale by absolute row through the descriptor's m_off)."""
    hip = _hip()
    a, w, b = rnd(M, K, seed=700 + K), rnd(N, K, seed=701, scale=0.05), rnd(N, seed=702)
    A, W = a.to(dt).cuda(), w.to(dt).cuda()
    if case == "strided":       # A is a column slice of a wider activation (row stride 1.5 K elements)
        wide = torch.zeros(M, K + K // 2, dtype=dt, device="cuda")
        wide[:, :K] = A
        A = wide[:, :K]
    ref = a.to(dt).double() @ w.to(dt).double().T
    kw, post = {}, (lambda x: x)
    if case in ("bias_rowscale", "f32res", "gelu", "gelu_save", "mul_saved", "dropout"):
        kw["bias"] = b.cuda()
        ref = ref + b.double()
    if case == "bias_rowscale":
        rs = (torch.arange(M // 8) % 3).float() * 0.5
        kw.update(row_scale=rs.cuda(), row_scale_group=8)
        ref = ref * rs.double().repeat_interleave(8)[:, None]
    if case == "f32res_rowscale":     # the ViT's fc2 in training: fp32 residual stream, drop-path scale per clip of 1569 token rows, ragged M
        kw["bias"] = b.cuda()
        rs = ((torch.arange(M // 1569) % 4) != 1).float() / 0.75
        kw.update(row_scale=rs.cuda(), row_scale_group=1569)
        ref = (ref + b.double()) * rs.double().repeat_interleave(1569)[:, None]
    if case in ("f32res", "f32res_rowscale"):
        res = rnd(M, N, seed=703)
        kw.update(residual=res.cuda(), out_dtype=torch.float32)
        ref = ref + res.double()
    saved = None
    if case == "gelu":
        kw["act"] = hip.ACT_GELU
        ref = gelu(ref)
    if case == "gelu_save":
        saved = torch.empty(M, N, dtype=dt, device="cuda")
        kw.update(act=hip.ACT_GELU_SAVE_GRAD, pre_act=saved)
    if case == "mul_saved":
        fac = rnd(M, N, seed=704).to(dt)
        kw.update(act=hip.ACT_MUL_SAVED, pre_act=fac.cuda())
        ref = ref * fac.double()
    if case == "dropout":
        kw.update(drop_p=0.1, drop_seed=4242)
    outs = {}
    for kind in (0, 1):
        with hip.option("gemm_kind", kind):
            if saved is not None:
                saved.zero_()
            guard = torch.full((M + 256, N), 7.0, dtype=kw.get("out_dtype", dt), device="cuda")
            outs[kind] = hip.gemm(A, W, out=guard[:M], **kw)
            assert bool((guard[M:] == 7.0).all()), "kind %d wrote rows behind M" % kind
            if saved is not None:
                outs[("saved", kind)] = saved.clone()
    # round 5: the tile walk does not touch results -- the static round-robin walk (gemm_sched 0) and repeated launches of the dynamic one (tickets
    # from per-XCD counters that the last workgroup of every launch hands back zeroed) are bitwise equal
    with hip.option("gemm_sched", 0):
        if saved is not None:
            saved.zero_()
        assert torch.equal(hip.gemm(A, W, **kw), outs[1]), "static walk differs"
        assert saved is None or torch.equal(saved, outs[("saved", 1)])
    for _ in range(2):
        assert torch.equal(hip.gemm(A, W, **kw), outs[1]), "dynamic walk: a repeated launch differs"
    tol = (2e-5, 2e-4 * math.sqrt(K / 768)) if case.startswith("f32res") else OUT_TOL[dt]
    if case == "gelu_save":
        x = ref.clone().requires_grad_(True)
        y = torch.nn.functional.gelu(x)
        y.sum().backward()
        close(outs[1], y.detach(), *tol, "gelu out")
        close(outs[("saved", 1)], x.grad, *tol, "saved gelu'")
        close(outs[("saved", 1)], outs[("saved", 0)].double().cpu(), *tol, "saved gelu' vs round-3 kernel")
    elif case == "dropout":
        kept = outs[1] != 0
        assert torch.equal(kept, outs[0] != 0) and 0.05 < float((~kept).float().mean()) < 0.15       # same hash mask as the round-3 kernel
        close(torch.where(kept, outs[1].float(), torch.zeros_like(outs[1], dtype=torch.float32)), torch.where(kept.cpu(), ref / 0.9, torch.zeros_like(ref)), *tol, "dropout")
    else:
        close(outs[1], ref, *tol, "8-phase kernel (%s)" % case)
    close(outs[1], outs[0].double().cpu(), tol[0] * 2, tol[1] * 2, "8-phase vs round-3 kernel")


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [14 * 256, 14 * 256 + 64, 41 * 256 + 144])
def test_gemm_saved_gelu_grad_in_the_tile_layout(dt, M):
    """Round 5: the gelu' a GELU Linear keeps for its backward may live in the 8-phase kernel's tile layout (desc.c2_tiled; buffer of
    alpro_gemm_c2_tiled_rows rows) -- written by GELU_SAVE_GRAD, read back by MUL_SAVED of the same (M, N), never through the LDS.  Both
    outputs (gelu(..) of the forward, dX of the backward) must equal the row-layout pair bit for bit, ragged last tile row included; shapes
    the kernel does not take answer 0 rows and a tiled descriptor for them is refused."""
    hip = _hip()
    N, K = 3072, 768
    a, w1, b1 = rnd(M, K, seed=910), rnd(N, K, seed=911, scale=0.05), rnd(N, seed=912)
    dy, w2t = rnd(M, K, seed=913), rnd(N, K, seed=914, scale=0.05)          # fc2's dgrad: dX1 (M, N) = dY (M, K) @ W2 (K, N) = dY @ (W2^T as (N, K))^T
    A, W1, DY, W2T = a.to(dt).cuda(), w1.to(dt).cuda(), dy.to(dt).cuda(), w2t.to(dt).cuda()
    B1 = b1.cuda()
    rows = hip.gemm_c2_tiled_rows(M, N, K, dt)
    assert rows == (M + 255) // 256 * 256
    u_rows = torch.empty(M, N, dtype=dt, device="cuda")
    y_rows = hip.gemm(A, W1, bias=B1, act=hip.ACT_GELU_SAVE_GRAD, pre_act=u_rows)
    dx_rows = hip.gemm(DY, W2T, act=hip.ACT_MUL_SAVED, pre_act=u_rows)
    u_tile = torch.full((rows, N), float("nan"), dtype=dt, device="cuda")
    guard = torch.full((M + 256, N), 7.0, dtype=dt, device="cuda")       # rows behind M of the row-layout outputs stay untouched
    y_tile = hip.gemm(A, W1, out=guard[:M], bias=B1, act=hip.ACT_GELU_SAVE_GRAD, pre_act=u_tile, c2_tiled=True)
    assert torch.equal(y_tile, y_rows) and bool((guard[M:] == 7.0).all())
    dx_tile = hip.gemm(DY, W2T, act=hip.ACT_MUL_SAVED, pre_act=u_tile, c2_tiled=True)
    assert torch.equal(dx_tile, dx_rows)
    # the tile layout is a permutation of the rows layout inside every (wave, fragment row) block of 16 x 64 elements
    t0 = u_tile.view(-1)[:1024].float().sort().values
    r0 = u_rows[:16, :64].reshape(-1).float().sort().values
    assert torch.equal(t0, r0)
    # against fp64 of the rounded operands (the pair's semantics): dX = (dY W2) * gelu'(A W1^T + b1)
    pre = A[:256].double() @ W1.double().t() + B1.double()
    gp = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * 3.141592653589793) ** 0.5
    ref = (DY[:256].double() @ W2T.double().t()) * gp
    err = (dx_tile[:256].double() - ref).abs().max() / ref.abs().max()
    assert err < (2e-3 if dt == torch.float16 else 1.5e-2), err
    # not available: too few tiles for the 8-phase kernel / an fp32 GEMM -> 0 rows, and the launch refuses the flag
    assert hip.gemm_c2_tiled_rows(512, N, K, dt) == 0 and hip.gemm_c2_tiled_rows(M, N, K, torch.float32) == 0
    with pytest.raises(Exception):
        with hip.option("gemm_kind", 0):   # (the round-3 kernel has no tile layout)
            hip.gemm(A, W1, bias=B1, act=hip.ACT_GELU_SAVE_GRAD, pre_act=u_tile, c2_tiled=True)


def _cu_thief():
    """tools/cu_thief.hip (n resident workgroups with 64 KiB of LDS each: the footprint of a collective's channel kernels), built on the box."""
    import ctypes
    import subprocess
    so = "/tmp/libcu_thief_test.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "cu_thief.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.cu_thief_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("stolen", [8, 40])
def test_gemm_tile_scheduler_with_compute_units_taken(stolen):
    """The persistent 8-phase GEMM while another kernel holds CUs (a collective's channels during the overlapped gradient exchange,
    run_pretrain_sparse.py:432,601): `stolen` workgroups of tools/cu_thief.hip stay resident on a second stream, so that many of the GEMM's 256
    workgroups cannot start until somebody leaves.  With the dynamic walk the resident workgroups draw the displaced ones' tiles from the
    per-XCD counters (and from other XCDs' when a list runs dry: the thief's CUs are not spread evenly); results are bitwise those of the
    undisturbed launch, for both walks, launch after launch."""
    import time
    hip = _hip()
    dt = torch.float16
    M, N, K = 256 * 196 + 32, 768, 768          # the B = 32 projection: 591 tiles = 2.3 rounds of 256, ragged last tile row
    a, w, b = rnd(M, K, seed=910), rnd(N, K, seed=911, scale=0.05), rnd(N, seed=912)
    A, W, Bv = a.to(dt).cuda(), w.to(dt).cuda(), b.cuda()
    ref = hip.gemm(A, W, bias=Bv)
    torch.cuda.synchronize()
    lib = _cu_thief()
    side = torch.cuda.Stream()
    buf = torch.zeros(64 * 65536, device="cuda")
    started = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.cu_thief_launch(buf.data_ptr(), stolen, int(100e6 * 0.5), started.data_ptr(), side.cuda_stream)   # ~0.5 s of the 100 MHz wall clock
    assert rc == 0
    t0 = time.time()
    while int(started.item()) < stolen:
        assert time.time() - t0 < 10.0, "thief workgroups did not become resident"
        time.sleep(0.001)
    for sched in (1, 0, 1):
        with hip.option("gemm_sched", sched):
            for _ in range(3):
                assert torch.equal(hip.gemm(A, W, bias=Bv), ref), "gemm_sched %d under contention" % sched
    torch.cuda.synchronize()
    assert torch.equal(hip.gemm(A, W, bias=Bv), ref)      # and undisturbed again afterwards (every launch left its scheduler block clean)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,L,group,masked,p", [(6, 197, 3, False, 0.0), (4, 40, 1, True, 0.0), (8, 30, 1, True, 0.1), (2, 5, 2, False, 0.0), (16, 197, 8, False, 0.0),
                                                    (3, 256, 1, True, 0.0)])
def test_attn_cls_precise_query(dt, batch, L, group, masked, p):
    """The CLS query's attention in fp32 (round 4) against fp64, in both forms: alpro_attn_cls_fwd (stand-alone: q and the CLS token's own
    k / v unrounded from the fp32 side tensor, one row per `group` sequences; the other tokens' K / V from the 16-bit qkv tensor) and the
    cls_q / cls_out side path of alpro_attn_fwd (same launch as the full attention; every K / V row, the CLS token's included, from the 16-bit
    images in LDS).  With probability dropout: under the mask alpro_attn_fwd draws for query 0.  The regular output must not move."""
    hip = _hip()
    H = 12
    qkv = rnd(batch * L, 3 * H * 64, seed=300 + L).to(dt)
    cls = rnd(batch // group, 3 * H * 64, seed=301 + L)
    bias = None
    if masked:
        m = torch.ones(batch, L)
        for b in range(batch):
            m[b, L - 2 - (3 * b) % (L - 4):] = 0
        bias = (1.0 - m) * -10000.0
    seed = 12345 if p > 0 else 0
    kb = None if bias is None else bias.cuda()
    out = hip.attn_cls(qkv.cuda(), cls.cuda(), batch, L, H, 0.125, group=group, key_bias=kb, drop_p=p, drop_seed=seed)
    plain = hip.attn(qkv.cuda(), batch, L, H, 0.125, kb, drop_p=p, drop_seed=seed)
    full, fused = hip.attn(qkv.cuda(), batch, L, H, 0.125, kb, drop_p=p, drop_seed=seed, cls_q=cls.cuda(), cls_group=group)
    assert torch.equal(full, plain)
    keep = None
    if p > 0:   # the mask alpro_attn_fwd draws for query 0 of (b, h): drop_keep(seed, ((b*H + h)*L + 0)*L + key) (attention.hip; numpy restatement)
        from tests.test_hip_bwd_ops import _keep_mask
        keep = _keep_mask(seed, (batch * H * L) * L, p).view(batch, H, L, L)[:, :, 0].double()
        assert 0.02 < float(1.0 - keep.mean()) < 0.3
    for name, got, own_kv in (("stand-alone", out, True), ("fused", fused, False)):
        t = qkv.double().view(batch, L, 3, H, 64).clone()
        c = cls.double().view(batch // group, 3, H, 64).repeat_interleave(group, 0)
        qc = c[:, 0]
        if own_kv:
            t[:, 0] = c                                     # the CLS token's own k / v: unrounded too
        k, v = t[:, :, 1], t[:, :, 2]                        # (b, L, H, 64)
        sc = torch.einsum("bhd,blhd->bhl", qc, k) * 0.125
        if bias is not None:
            sc = sc + bias[:, None, :].double()
        pr = sc.softmax(-1)
        if keep is not None:
            pr = pr * keep / (1.0 - p)
        ref = torch.einsum("bhl,blhd->bhd", pr, v).reshape(batch, H * 64)
        close(got, ref, 2e-5, 2e-5, "attn_cls " + name)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(15168, 768, 768), (45 * 256 + 17, 1024, 128), (2560, 3072, 768)])
def test_gemm_persistent_partial_grid(M, N, K):
    """Persistent 256x256 kernel with fewer tiles than CUs (grid not a multiple of 8 before rounding): every tile is visited."""
    hip = _hip()
    dt = torch.bfloat16
    a, w = rnd(M, K, seed=40).to(dt), rnd(N, K, seed=41, scale=0.05).to(dt)
    with hip.option("gemm_tile", 256):
        out = hip.gemm(a.cuda(), w.cuda())
    ref = (a.cuda().float() @ w.cuda().float().T)
    err = (out.float() - ref).abs().max().item()
    assert err < 2e-2 * max(1.0, ref.abs().max().item()), err
