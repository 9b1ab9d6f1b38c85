"""Rounding-site model of the 16-bit operand modes (CPU, no GPU, no kernels): which roundings set the VTC-logit error?

    python tools/precision_model.py [--B 8] [--T 4] [--Lt 30] [--configs all]

A plain-torch fp32 restatement of the two encoders' forward (ViT divided space-time blocks as alpro_amd/modeling/timesformer/vit.py runs
them -- merged temporal projection, fp32 residual stream, 16-bit branch outputs -- and the six text-mode BERT layers as xbert.py runs them)
in which every place where the HIP path rounds a value to the operand dtype is an explicit `r(site, tensor)` call.  Switching a site class
off (or switching it off for the CLS rows only) and comparing the resulting VTC logits with the all-fp32 run says how much of the logit
error that class of roundings carries.  Accumulation order differs from the kernels', the rounding sites and their statistics do not; the
model is a design aid, not a parity check -- weights and inputs are the closed forms of tests/golden/det_init.py.
"""
import argparse
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.golden.det_init import det_batch, det_param  # noqa: E402

D, H, HD = 768, 12, 64
# what the precise CLS-row side path covers: "full" = LN -> qkv -> attention query row -> proj -> MLP all unrounded; "osave" = the attention kernel's
# fp32 accumulator row of the CLS query (16-bit q / k / v / P as usual, output row NOT rounded) -> proj -> MLP unrounded; "projmlp" = the rounded
# attention output row -> proj -> MLP unrounded; "mlp" = MLP only
CLS_VARIANT = ["full"]
CLS_W16 = [False]   # --cls-w16: the precise rows see the 16-bit ROUNDED weights (activations still unrounded): the cost question of DESIGN.md section 8
VIT_EPS, BERT_EPS = 1e-6, 1e-12


class Sites:
    """r(site, x): round x to the 16-bit dtype unless the site (or its class prefix) is switched off."""

    def __init__(self, dtype=torch.float16, off=()):
        self.dtype, self.off = dtype, set(off)

    def on(self, site):
        parts = site.split(".")
        return not any(".".join(parts[:i]) in self.off for i in range(1, len(parts) + 1)) and "all" not in self.off

    def __call__(self, site, x):
        return x.to(self.dtype).to(torch.float32) if self.on(site) else x


_PCACHE = {}


def P(name, shape):
    k = (name, tuple(shape))
    if k not in _PCACHE:
        _PCACHE[k] = det_param(name, shape)
    return _PCACHE[k]


def lin(r, site, x, w, b, out_round=True, rows32=None):
    """y = r(x) @ r(w)^T + b, output rounded at site.out; rows32: boolean row mask of rows computed WITHOUT any rounding (a precise side path)."""
    y = F.linear(r(site + ".in", x), r(site + ".w", w), b)
    if out_round:
        y = r(site + ".out", y)
    if rows32 is not None and rows32.any():
        y = y.clone()
        y[rows32] = F.linear(x[rows32], r(site + ".w", w) if CLS_W16[0] else w, b)
    return y


def attention(r, site, qkv, L, scale, key_bias=None, rows32=None):
    """qkv (Bq, L, 3*D) (already carrying the qkv GEMM's output rounding) -> (Bq, L, D); P rounded before PV, output rounded."""
    Bq = qkv.shape[0]
    q, k, v = qkv.view(Bq, L, 3, H, HD).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    m = s.max(-1, keepdim=True)[0]
    e = torch.exp(s - m)
    o = (r(site + ".p", e) @ v) / e.sum(-1, keepdim=True)
    o = r(site + ".out", o.transpose(1, 2).reshape(Bq, L, D))
    if rows32 is not None and rows32.any() and CLS_VARIANT[0] in ("full", "osave"):
        # precise query rows: "full" unrounded P too (K / V stay what the big GEMM produced); "osave": the kernel's own fp32 accumulator row
        pe = e if CLS_VARIANT[0] == "full" else r(site + ".p", e)
        o32 = ((pe @ v) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(Bq, L, D)
        o = torch.where(rows32.view(Bq, L, 1), o32, o)
    return o


def vit_forward(r, x, T, cls32_blocks=(), pre="visual_encoder.model"):
    """x (B, T, 3, 224, 224) -> CLS row of norm(blocks(tokens)) (B, D).  cls32_blocks: blocks whose CLS rows take a precise (fp32) side path
    in the spatial branch (LN1 -> qkv -> attention query -> proj) and the MLP."""
    B = x.shape[0]
    N = 196
    g = lambda n, *s: P(pre + "." + n, s)  # noqa: E731
    patches = x.reshape(B * T, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(B * T * N, 768)
    tokp = lin(r, "patch", patches, g("patch_embed.proj.weight", D, 3, 16, 16).view(D, -1), g("patch_embed.proj.bias", D), out_round=False)
    pos, tim = g("pos_embed", 1, N + 1, D), g("time_embed", 1, T, D)
    tokp = tokp.view(B, T, N, D) + pos[0, 1:][None, None] + tim[0][None, :, None]
    tok = tokp.permute(0, 2, 1, 3).reshape(B, N * T, D)                     # (n, t) order
    cls = (g("cls_token", 1, 1, D) + pos[:, :1]).expand(B, 1, D)
    xs = torch.cat([cls, tok], 1)
    scale = HD ** -0.5
    for i in range(12):
        b = "blocks.%d." % i
        w = lambda n, *s: g(b + n, *s)  # noqa: E731
        ln = lambda t, n: F.layer_norm(t, (D,), w(n + ".weight", D), w(n + ".bias", D), VIT_EPS)  # noqa: E731
        c32 = i in cls32_blocks
        # temporal (patch rows only)
        xt = xs[:, 1:].reshape(B * N, T, D)
        qkv = lin(r, "t_qkv", ln(xt, "temporal_norm1"), w("temporal_attn.qkv.weight", 3 * D, D), w("temporal_attn.qkv.bias", 3 * D))
        a = attention(r, "t_attn", qkv, T, scale)
        wfc, wp, bp = w("temporal_fc.weight", D, D), w("temporal_attn.proj.weight", D, D), w("temporal_attn.proj.bias", D)
        d_t = lin(r, "t_proj", a, wfc @ wp, wfc @ bp)                          # merged projection, 16-bit delta
        xs = torch.cat([xs[:, :1], xs[:, 1:] + d_t.view(B, N * T, D) + w("temporal_fc.bias", D)], 1)
        # spatial: frame-token order (B*T, 1+N), CLS replicated
        pt = xs[:, 1:].view(B, N, T, D).permute(0, 2, 1, 3).reshape(B * T, N, D)
        fr = torch.cat([xs[:, :1].expand(B, T, D).reshape(B * T, 1, D), pt], 1)
        rows_cls = torch.zeros(B * T, N + 1, dtype=torch.bool)
        rows_cls[:, 0] = c32
        rc = rows_cls.view(-1) if c32 else None
        qkv = lin(r, "s_qkv", ln(fr, "norm1").view(-1, D), w("attn.qkv.weight", 3 * D, D), w("attn.qkv.bias", 3 * D), rows32=rc if CLS_VARIANT[0] == "full" else None).view(B * T, N + 1, 3 * D)
        a = attention(r, "s_attn", qkv, N + 1, scale, rows32=rows_cls if c32 else None)
        d_s = lin(r, "s_proj", a.view(-1, D), w("attn.proj.weight", D, D), w("attn.proj.bias", D), rows32=rc if CLS_VARIANT[0] != "mlp" else None).view(B, T, N + 1, D)
        cls_new = xs[:, 0] + d_s[:, :, 0].mean(1)
        pt_new = xs[:, 1:] + d_s[:, :, 1:].permute(0, 2, 1, 3).reshape(B, N * T, D)
        xs = torch.cat([cls_new[:, None], pt_new], 1)
        # MLP
        rows_m = torch.zeros(B, 1 + N * T, dtype=torch.bool)
        rows_m[:, 0] = c32
        rm = rows_m.view(-1) if c32 else None
        f1 = lin(r, "fc1", ln(xs, "norm2").view(-1, D), w("mlp.fc1.weight", 4 * D, D), w("mlp.fc1.bias", 4 * D), out_round=False)
        g1 = 0.5 * f1 * (1.0 + torch.erf(f1 / math.sqrt(2.0)))
        g1r = r("fc1.out", g1)
        if rm is not None:
            g1r = g1r.clone()
            g1r[rm] = F.gelu(F.linear(F.layer_norm(xs, (D,), w("norm2.weight", D), w("norm2.bias", D), VIT_EPS).view(-1, D)[rm], w("mlp.fc1.weight", 4 * D, D), w("mlp.fc1.bias", 4 * D)))
        xs = xs + lin(r, "fc2", g1r, w("mlp.fc2.weight", D, 4 * D), w("mlp.fc2.bias", D), out_round=False, rows32=rm).view(B, 1 + N * T, D)
    return F.layer_norm(xs[:, 0], (D,), g("norm.weight", D), g("norm.bias", D), VIT_EPS)


def text_forward(r, ids, mask, cls32_layers=(), pre="text_encoder.bert"):
    """(B, L) ids / mask -> CLS row of the text-mode encoder output (layers 0..5), (B, D)."""
    B, L = ids.shape
    g = lambda n, *s: P(pre + "." + n, s)  # noqa: E731
    e = g("embeddings.word_embeddings.weight", 30522, D)[ids] + g("embeddings.position_embeddings.weight", 512, D)[:L][None] + g("embeddings.token_type_embeddings.weight", 2, D)[0]
    h32 = F.layer_norm(e, (D,), g("embeddings.LayerNorm.weight", D), g("embeddings.LayerNorm.bias", D), BERT_EPS).view(B * L, D)
    kb = (1.0 - mask.float()) * -10000.0
    scale = 1.0 / math.sqrt(HD)
    for i in range(6):
        lay = "encoder.layer.%d." % i
        w = lambda n, *s: g(lay + n, *s)  # noqa: E731
        c32 = i in cls32_layers
        rows = torch.zeros(B, L, dtype=torch.bool)
        rows[:, 0] = c32
        rc = rows.view(-1) if c32 else None
        wqkv = torch.cat([w("attention.self.%s.weight" % n, D, D) for n in ("query", "key", "value")], 0)
        bqkv = torch.cat([w("attention.self.%s.bias" % n, D) for n in ("query", "key", "value")], 0)
        qkv = lin(r, "b_qkv", h32, wqkv, bqkv, rows32=rc if CLS_VARIANT[0] == "full" else None).view(B, L, 3 * D)
        ctx = attention(r, "b_attn", qkv, L, scale, key_bias=kb, rows32=rows if c32 else None).view(-1, D)
        d1 = lin(r, "b_ao", ctx, w("attention.output.dense.weight", D, D), w("attention.output.dense.bias", D), rows32=rc if CLS_VARIANT[0] != "mlp" else None)
        a32 = F.layer_norm(h32 + d1, (D,), w("attention.output.LayerNorm.weight", D), w("attention.output.LayerNorm.bias", D), BERT_EPS)
        f1 = lin(r, "b_i", a32, w("intermediate.dense.weight", 4 * D, D), w("intermediate.dense.bias", 4 * D), out_round=False)
        it = r("b_i.out", F.gelu(f1))
        if rc is not None:
            it = it.clone()
            it[rc] = F.gelu(F.linear(a32[rc], w("intermediate.dense.weight", 4 * D, D), w("intermediate.dense.bias", 4 * D)))
        d2 = lin(r, "b_o", it, w("output.dense.weight", D, 4 * D), w("output.dense.bias", D), rows32=rc)
        h32 = F.layer_norm(a32 + d2, (D,), w("output.LayerNorm.weight", D), w("output.LayerNorm.bias", D), BERT_EPS)
    return h32.view(B, L, D)[:, 0]


def feats(r, batch, T, vis_cls32=(), txt_cls32=()):
    v = vit_forward(r, batch["visual_inputs"], T, vis_cls32)
    t = text_forward(r, batch["text_input_ids"], batch["text_input_mask"], txt_cls32)
    vf = F.normalize(F.linear(v, P("vision_proj.weight", (256, D)), P("vision_proj.bias", (256,))), dim=-1)
    tf = F.normalize(F.linear(t, P("text_proj.weight", (256, D)), P("text_proj.bias", (256,))), dim=-1)
    return vf, tf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=6)
    ap.add_argument("--T", type=int, default=4)
    ap.add_argument("--Lt", type=int, default=30)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--seed-name", default="precision_model")
    ap.add_argument("--only", default="")
    ap.add_argument("--cls-variant", default="full", choices=["full", "osave", "projmlp", "mlp"])
    ap.add_argument("--cls-w16", type=int, default=0)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    CLS_VARIANT[0] = args.cls_variant
    CLS_W16[0] = bool(args.cls_w16)
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    batch = det_batch(args.B, args.T, Lt=args.Lt, seed_name=args.seed_name, with_mlm=False, with_mpm=False)
    all12, all6 = tuple(range(12)), tuple(range(6))
    vis_sites = ["patch", "t_qkv", "t_attn", "t_proj", "s_qkv", "s_attn", "s_proj", "fc1", "fc2"]
    txt_sites = ["b_qkv", "b_attn", "b_ao", "b_i", "b_o"]
    configs = [("fp32 (reference)", dict(off=["all"])),
               ("%s plain" % args.dtype, dict()),
               ("visual 16-bit, text fp32", dict(off=txt_sites)),
               ("visual fp32, text 16-bit", dict(off=vis_sites)),
               ("CLS rows precise: ViT all blocks", dict(vis=all12)),
               ("CLS rows precise: ViT all blocks + text all layers", dict(vis=all12, txt=all6)),
               ("CLS rows precise: ViT last 4 + text last 2", dict(vis=(8, 9, 10, 11), txt=(4, 5))),
               ("CLS rows precise: ViT last 2 + text last 1", dict(vis=(10, 11), txt=(5,))),
               ("CLS rows precise: text all layers only", dict(txt=all6)),
               ("no weight rounding anywhere", dict(off=[s + ".w" for s in vis_sites + txt_sites])),
               ("no output (delta / activation) rounding anywhere", dict(off=[s + ".out" for s in vis_sites + txt_sites])),
               ("no A-operand (LayerNorm output) rounding anywhere", dict(off=[s + ".in" for s in vis_sites + txt_sites])),
               ("no P rounding in attention", dict(off=["t_attn.p", "s_attn.p", "b_attn.p"]))]
    for s in vis_sites:
        configs.append(("visual site class off: " + s, dict(off=[s])))
    for s in txt_sites:
        configs.append(("text site class off: " + s, dict(off=[s])))
    if args.only:
        configs = [c for c in configs if c[0].startswith("fp32") or args.only in c[0]]
    ref = None
    print("B=%d T=%d Lt=%d dtype=%s: %d x %d logits per config" % (args.B, args.T, args.Lt, args.dtype, args.B, args.B))
    for name, c in configs:
        t0 = time.time()
        r = Sites(dt, c.get("off", ()))
        with torch.no_grad():
            vf, tf = feats(r, batch, args.T, c.get("vis", ()), c.get("txt", ()))
        if ref is None:
            ref = (vf, tf)
            print("%-62s (%.0f s)" % (name, time.time() - t0))
            continue
        temp = 0.07
        e_all = ((vf @ tf.t()) - (ref[0] @ ref[1].t())).abs() / temp
        e_v = ((vf @ ref[1].t()) - (ref[0] @ ref[1].t())).abs() / temp
        e_t = ((ref[0] @ tf.t()) - (ref[0] @ ref[1].t())).abs() / temp
        print("%-62s logits max %.2e rms %.2e | video-side max %.2e rms %.2e | text-side max %.2e rms %.2e  (%.0f s)" % (
            name, e_all.max(), e_all.pow(2).mean().sqrt(), e_v.max(), e_v.pow(2).mean().sqrt(), e_t.max(), e_t.pow(2).mean().sqrt(), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
