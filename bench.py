"""Benchmark of the ALPRO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload pretrain_step|visual_fwd|pretrain_fwd] [--batch B] [--dtype fp16|bf16|fp32]

Default mode: fp16 operands (fp32 accumulation / residual stream / statistics, dynamic loss scaling for the backward) + the CLS rows of both
encoders re-evaluated in fp32 (alpro_amd.config.cls_precise, round 4) -- the fastest mode that meets the north star's "VTC logits within
1e-3 of the reference" on EVERY reference fixture (plain fp16: 1.06e-3 on the worst one; bf16: 9e-3).  The `parity` object of the JSON line
is MEASURED in this process against all four reference-generated VTC fixtures under tests/golden/ (measure_parity below): `meets_bar` is
decided by the WORST of them.

Default workload: pretrain_step -- the configuration BASELINE.json's metric ("video-text pairs/sec at 1/2/4/8 MI355X") is
quoted on (configs[2] on one GPU, configs[3] under DP); visual_fwd is configs[1] (encoder-only isolation run).

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU over RCCL).  Rank 0 prints ONE JSON line.  A "step" is one pass of the hot path over one
synthetic batch already resident in HBM; weights are random-init of the real architecture.

Workloads (BASELINE.json configs):
  visual_fwd   configs[1]: TimeSformer-divST visual encoder only, B=32 clips x 8 frames x 224^2, bf16 forward
  pretrain_fwd configs[2] forward half: AlproForPretrain VTC+VTM+MLM+MPM forward, B=64 pairs (eval-mode dropout)
  pretrain_step configs[2]/[3]: full training step of AlproForPretrain (VTC+VTM+MLM+MPM forward, hand-written HIP
               backward, gradient all-reduce over RCCL, grad-norm clip 20.0 + AdamW lr 1e-4 betas (0.9, 0.98) as
               config_release/pretrain_alpro.json), B pairs per GPU, drop_path 0.1 active
The forward-only workloads shard by clip with no collective; pretrain_step adds the in-forward feature all-gather and
the gradient all-reduce (weak scaling: every rank runs B pairs).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

BERT_CFG = {"attention_probs_dropout_prob": 0.1, "hidden_act": "gelu", "hidden_dropout_prob": 0.1, "hidden_size": 768,
            "initializer_range": 0.02, "intermediate_size": 3072, "layer_norm_eps": 1e-12, "max_position_embeddings": 512,
            "model_type": "bert", "num_attention_heads": 12, "num_hidden_layers": 12, "pad_token_id": 0,
            "type_vocab_size": 2, "vocab_size": 30522, "fusion_layer": 6, "encoder_width": 768, "itc_token_type": "cls"}
VENC = {"cls": "TimeSformer", "patch_size": 16, "attn_drop_rate": 0, "drop_rate": 0, "drop_path_rate": 0.1,
        "maxpool_kernel_size": 2, "use_maxpooling": False, "gradient_checkpointing": False, "img_size": 224}
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense, MI355X_MICROARCH.md
VISUAL_GFLOP_PER_CLIP_8F = 391.7  # 12 x 32.48 + 1.85 (2*M*N*K, forward)


class Cfg:
    def __init__(self, d):
        self.__dict__.update(d)
        self.num_entities = 1000
        self.max_n_example_per_group = 1


def synth_batch(B, T, device, seed, full):
    g = torch.Generator(device="cpu").manual_seed(seed)
    batch = {"visual_inputs": torch.randn(B, T, 3, 224, 224, generator=g).to(device)}
    ids = torch.randint(1000, 30000, (B, 40), generator=g)
    ids[:, 0] = 101
    batch["text_input_ids"] = ids.to(device)
    batch["text_input_mask"] = torch.ones(B, 40, dtype=torch.long, device=device)
    if full:
        sel = torch.rand(B, 40, generator=g) < 0.15
        sel[:, 0] = False
        sel[:, 1] = True
        mlm = ids.clone()
        mlm[sel] = 103
        lab = torch.full((B, 40), -100, dtype=torch.long)
        lab[sel] = ids[sel]
        batch["mlm_text_input_ids"], batch["mlm_labels"] = mlm.to(device), lab.to(device)
        mask = torch.ones(B, 14, 14)
        mask[:, 4:10, 4:10] = 0
        batch["mpm_mask"] = mask.to(device)
        batch["crop_visual_inputs"] = torch.randn(B, T, 3, 224, 224, generator=g).to(device)
        batch["context_visual_inputs"] = batch["visual_inputs"]
        batch["type"] = "video"
    return batch


class KernelTimer:
    """Per-kernel-family device time via HIP events on the launch stream (a separate, untimed pass)."""

    def __init__(self, hip):
        self.hip, self.rec, self.orig = hip, [], {}

    def __enter__(self):
        def wrap(name, flops_fn, key_fn=None):
            fn = getattr(self.hip, name)
            self.orig[name] = fn

            def timed(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **k)
                e1.record()
                self.rec.append((name, flops_fn(*a, **k), e0, e1, key_fn(*a, **k) if key_fn else ""))
                return out
            setattr(self.hip, name, timed)

        def gemm_key(a, w, *r, **k):
            od = k.get("out_dtype")
            return "M=%d N=%d K=%d act=%s res=%d out=%s map=%s" % (a.shape[0], w.shape[0], a.shape[1], k.get("act", 0), k.get("residual") is not None,
                                                              str(od).replace("torch.", "") if od is not None else "-", k.get("map_mode", 0))
        wrap("gemm", lambda a, w, *r, **k: 2.0 * a.shape[0] * a.shape[1] * w.shape[0], gemm_key)
        # round 6: the temporal half's qkv Linear + frame attention in one launch: priced on the GEMM's flops plus the attention's (the same two
        # figures the separate launches are priced on); its time joins the GEMM family of the roofline object, where the qkv GEMM's was
        wrap("gemm_qkv_tattn", lambda a, w, bias, T, H, *r, **k: 2.0 * a.shape[0] * a.shape[1] * w.shape[0] + 4.0 * a.shape[0] * T * H * 64,
             lambda a, w, bias, T, H, *r, **k: "M=%d N=%d K=%d T=%d fused qkv + temporal attention" % (a.shape[0], w.shape[0], a.shape[1], T))
        wrap("attn", lambda qkv, batch, L, H, *r, **k: 4.0 * batch * H * L * L * 64)
        wrap("attn_temporal", lambda qkv, T, H, *r, **k: 4.0 * qkv.shape[0] * T * H * 64)
        wrap("layernorm", lambda *a, **k: 0.0)
        wrap("add_layernorm", lambda *a, **k: 0.0)
        wrap("add_layernorm_pre_mlp2", lambda *a, **k: 0.0)
        wrap("gemm_rows", lambda *a, **k: 0.0)
        wrap("attn_bwd", lambda qkv, out, dout, lse, batch, L, H, *r, **k: 10.0 * batch * H * L * L * 64)
        wrap("attn_temporal_bwd", lambda qkv, out, dout, lse, T, H, *r, **k: 10.0 * qkv.shape[0] * T * H * 64)
        wrap("layernorm_bwd", lambda *a, **k: 0.0)
        wrap("transpose", lambda *a, **k: 0.0)
        wrap("gather_cast", lambda *a, **k: 0.0)
        wrap("gelu_bwd", lambda *a, **k: 0.0)
        wrap("gemm_tn_acc", lambda a, b, c, *r, **k: 2.0 * a.shape[0] * a.shape[1] * b.shape[1],
             lambda a, b, c, *r, **k: "M=%d N=%d K=%d" % (a.shape[0], a.shape[1], b.shape[1]))
        wrap("colsum_acc", lambda *a, **k: 0.0)
        wrap("adamw_step", lambda *a, **k: 0.0)
        wrap("sumsq", lambda *a, **k: 0.0)
        return self

    def __exit__(self, *a):
        for n, f in self.orig.items():
            setattr(self.hip, n, f)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        shapes = {}
        for name, fl, e0, e1, key in self.rec:
            d = agg.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl
            d[2] += e0.elapsed_time(e1)
            if key:
                sd = shapes.setdefault((name, key), [0, 0.0, 0.0])
                sd[0] += 1
                sd[1] += fl
                sd[2] += e0.elapsed_time(e1)
        self.shapes = shapes
        if os.environ.get("ALPRO_BENCH_SHAPES"):  # per-shape GEMM table (tuning aid), to stderr
            for (name, key), (c, f, ms) in sorted(shapes.items(), key=lambda kv: -kv[1][2]):
                sys.stderr.write("%-12s %-70s n=%3d  %8.3f ms  %6.0f TF/s\n" % (name, key, c, ms, f / ms / 1e9))
        return {n: {"launches": c, "flops": f, "ms": ms} for n, (c, f, ms) in agg.items()}


def cpu_baseline_train(T):
    """The oracle's full pretraining step (VTC+VTM+MLM+MPM forward + autograd backward, fp32) on the host cores: the faster of TWO
    steps of B=2 pairs after a warm-up forward+backward (thread pool, allocator, autograd graph caches) -- a bounded sample
    (~20-30 s of CPU work)."""
    from oracle import alpro_oracle as ao
    from tests.golden.det_init import det_batch
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    spec = ao.alpro_state_spec("pretrain", BERT_CFG, T)
    skip = ("prompter.text_encoder.", "prompter.itm_head", "prompter.text_proj", "visual_encoder.model.head", "prompter.visual_encoder.model.head")
    g = torch.Generator().manual_seed(0)
    p = {}
    for k, s in spec.items():
        if k.startswith(skip) or "decoder" in k:
            continue
        if k.endswith("position_ids"):
            p[k] = torch.arange(s[1]).view(1, -1)
        elif k.endswith(("norm1.weight", "norm2.weight", "norm.weight", "LayerNorm.weight")):
            p[k] = torch.ones(*s)
        elif k.endswith("temp"):
            p[k] = torch.tensor(0.07)
        else:
            p[k] = torch.randn(*s, generator=g) * 0.02
    for k, v in p.items():
        if v.is_floating_point() and not k.startswith("prompter."):
            v.requires_grad_(True)
    orc = ao.AlproOracle(p, BERT_CFG, T)
    Bc = 2
    batch = det_batch(Bc, T, seed_name="cpu_baseline")
    def one_step(b):
        for v in p.values():
            v.grad = None
        t0 = time.time()
        out = orc.forward_pretrain(b)
        (out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]).backward()
        return time.time() - t0
    warm = {k: (v[:1] if torch.is_tensor(v) else v) for k, v in batch.items()}
    one_step(warm)                                        # warm-up: one pair forward + backward
    times = [one_step(batch), one_step(batch)]
    dt = min(times)
    return {"value": Bc / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": "oracle AlproForPretrain fwd+bwd (no optimizer), min of 2 steps of %d pairs x %df x 224^2 + 40 tok after a warm-up step, fp32, torch %d threads, %.1f / %.1f s"
                      % (Bc, T, threads, times[0], times[1])}


def cpu_baseline(T, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference, fp32) timed on this box's host cores: visual encoder forward."""
    from oracle import alpro_oracle as ao
    spec = ao.alpro_state_spec("retrieval", BERT_CFG, T)
    g = torch.Generator().manual_seed(0)
    p = {k: torch.randn(*s, generator=g) * 0.02 for k, s in spec.items() if k.startswith("visual_encoder") and "head" not in k}
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    Bc = 2
    x = torch.randn(Bc, 3, T, 224, 224, generator=g)
    with torch.no_grad():
        ao.timesformer_forward_features(x, p, "visual_encoder", T)  # warm-up
        t0, n = time.time(), 0
        while n < 1 or (time.time() - t0 < seconds_budget and n < 8):
            ao.timesformer_forward_features(x, p, "visual_encoder", T)
            n += 1
        dt = (time.time() - t0) / n
    return {"value": Bc / dt, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "oracle TimeSformer forward, %d clips x %df x 224^2 fp32, %d passes, torch %d threads" % (Bc, T, n, threads)}


DIVST_GFLOP_PER_CLIP_8F = 212.1  # LN + qkv + attention + proj (+ temporal_fc) of both halves, 12 blocks, 2*M*N*K by the reference's count


def measure_divst(dev, T, model=None, B=32, iters=5):
    """BASELINE configs[1] / north_star target: the divided space-time attention sub-blocks (vit.py:146-196: temporal LN + qkv +
    attention + proj/temporal_fc, spatial LN + qkv + attention + proj + CLS mean) of a B=32 x 8f forward, timed with HIP events on the
    launch stream: from Block.forward's entry to the return of alpro_cls_mean_residual (the last kernel before the MLP half).
    Priced against the dense bf16 MFMA peak on the reference's FLOP count (212.1 GFLOP per 8-frame clip; the merged temporal projection
    executes fewer)."""
    from alpro_amd import config as rt
    from alpro_amd import hip
    from alpro_amd.modeling.timesformer import vit
    if model is None:
        model = vit.TimeSformer(dict(VENC, num_frm=T), input_format="RGB").eval().to(dev)
    x = torch.randn(B, 3, T, 224, 224, device=dev)
    marks, tails = [], []
    deferred = rt.defer_temporal_add()   # round 6: the add + norm2 kernel also carries the temporal branch's add (10.5 KB per row, 1.5 of them the MLP half's norm2 rows)
    mlp_share = 1.0 / 7.0 if deferred else 1.0 / 6.0
    orig_fwd, orig_cls, orig_add, orig_add2 = vit.Block.forward, hip.cls_mean_residual, hip.add_layernorm, hip.add_layernorm_pre_mlp2

    def fwd(self, *a, **k):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        marks.append([e0, None])
        return orig_fwd(self, *a, **k)

    def cls(*a, **k):      # round-2 form (Block.fuse_residual_ln = False): the sub-blocks end with alpro_cls_mean_residual
        out = orig_cls(*a, **k)
        if marks[-1][1] is None:   # (the precise-CLS chain calls the same kernel AFTER the MLP half, round 4: only the first end mark of a block counts)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            marks[-1][1] = e1
        return out

    def end_mark(fn, a, k):
        ea = torch.cuda.Event(enable_timing=True)
        ea.record()
        out = fn(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        marks[-1][1] = e1
        tails.append((ea, e1))
        return out

    def add_ln(*a, **k):   # round-3 form: they end inside the PRE_MLP add-LayerNorm kernel (see the accounting note below)
        if k.get("mode") != hip.ADD_PRE_MLP:
            return orig_add(*a, **k)
        return end_mark(orig_add, a, k)

    def add_ln2(*a, **k):  # round 6: the same kernel with the temporal branch's deferred add (alpro_add_layernorm_pre_mlp2)
        return end_mark(orig_add2, a, k)
    # The region is timed with HIP events on the MAIN stream, so it is measured with the precise-CLS chain on that stream too (ALPRO_CLS_STREAM=0 for
    # this pass): every launch the sub-blocks need is then inside the region.  On its side stream (the inference default) part of the chain runs
    # beside the MLP half and the region would both lose that work and gain the main stream's waits (profiles/r5_cls_stream_ab.txt).
    # ... and, for the same reason, with the whole batch on the launch stream (ALPRO_SPLIT_STREAMS off for this pass: the product default runs the
    # two halves of the batch on two free-running streams, profiles/r6_split_streams_ab.txt, where a block's region has no single start and end)
    prev_cs, prev_split = rt._cls_stream[0], rt._split_streams[0]
    rt.set_cls_stream("0")
    rt.set_split_streams("0")
    with torch.no_grad():
        for _ in range(2):
            model.forward_features(x)
        vit.Block.forward, hip.cls_mean_residual, hip.add_layernorm, hip.add_layernorm_pre_mlp2 = fwd, cls, add_ln, add_ln2
        try:
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(iters):
                model.forward_features(x)
            t1.record()
            torch.cuda.synchronize()
        finally:
            vit.Block.forward, hip.cls_mean_residual, hip.add_layernorm, hip.add_layernorm_pre_mlp2 = orig_fwd, orig_cls, orig_add, orig_add2
            rt.set_cls_stream(prev_cs)
            rt.set_split_streams(prev_split)
    # Per-block table: one more forward with every library call of the region under its own pair of HIP events; the rows are
    # means over the 12 blocks, `unattributed` is what the region's own events measured beyond the sum (launch gaps, torch's small copies).
    table = None
    try:
        spans = []
        with torch.no_grad(), KernelTimer(hip) as kt2:
            rec = kt2.rec
            wrapped_add = hip.add_layernorm

            def fwd2(self, *a, **k):
                spans.append([len(rec), None])
                return orig_fwd(self, *a, **k)

            wrapped_add_p2 = hip.add_layernorm_pre_mlp2

            def add2(*a, **k):
                out = wrapped_add(*a, **k)
                if k.get("mode") == hip.ADD_PRE_MLP:
                    spans[-1][1] = len(rec)
                return out

            def add2_p2(*a, **k):
                out = wrapped_add_p2(*a, **k)
                spans[-1][1] = len(rec)
                return out
            vit.Block.forward, hip.add_layernorm, hip.add_layernorm_pre_mlp2 = fwd2, add2, add2_p2
            rt.set_cls_stream("0")
            rt.set_split_streams("0")
            try:
                model.forward_features(x)
            finally:
                vit.Block.forward, hip.add_layernorm, hip.add_layernorm_pre_mlp2 = orig_fwd, wrapped_add, wrapped_add_p2
                rt.set_cls_stream(prev_cs)
                rt.set_split_streams(prev_split)
        torch.cuda.synchronize()
        rows = {}
        for lo, hi in spans:
            if hi is None:
                continue
            seq = rec[lo:hi]
            n_gemm = 0
            for i, (name, fl, e0, e1, key) in enumerate(seq):
                us = e0.elapsed_time(e1) * 1e3
                if name == "gemm":
                    n_gemm += 1
                    fused = any(r[0] == "gemm_qkv_tattn" for r in seq)
                    order = ["temporal projection (merged proj + temporal_fc) GEMM", "spatial qkv GEMM", "spatial projection GEMM"] if fused else \
                            ["temporal qkv GEMM", "temporal projection (merged proj + temporal_fc) GEMM", "spatial qkv GEMM", "spatial projection GEMM"]
                    label = order[n_gemm - 1] if n_gemm <= len(order) else "other GEMM"
                elif name == "add_layernorm_pre_mlp2":
                    label = "residual adds (temporal + spatial) + norm2 (6/7 counted)"
                    us *= 6.0 / 7.0
                elif name == "add_layernorm":
                    last = i == len(seq) - 1
                    label = "residual add + norm2 (5/6 counted)" if last else ("residual add + norm1" if not deferred else "norm1 of block input + temporal branch (nothing but the normalised rows written)")
                    if last:
                        us *= 5.0 / 6.0
                else:
                    label = {"layernorm": "temporal_norm1", "gemm_qkv_tattn": "temporal qkv GEMM + frame attention (fused)", "attn_temporal": "temporal attention",
                             "attn": "spatial attention (+ precise CLS query)", "gemm_rows": "precise CLS rows (fp32 q | k | v of the CLS token)"}.get(name, name)
                d_ = rows.setdefault(label, [0, 0.0])
                d_[0] += 1
                d_[1] += us
        nb = max(1, len([1 for lo, hi in spans if hi is not None]))
        table = {k_: round(v_[1] / nb, 1) for k_, v_ in rows.items()}
        table["sum_us_per_block"] = round(sum(table.values()), 1)
    except Exception as e:   # the table is a diagnostic: never let it take the measurement down
        table = {"error": repr(e)}
    # The PRE_MLP kernel is the spatial half's residual add (reads x and the 16-bit delta, writes x': 7.5 of its 9 KB per row) AND the MLP
    # half's norm2 (the 16-bit normalised row: 1.5 KB).  Its time is attributed 5/6 to the attention sub-blocks, 1/6 to the MLP.
    tail_ms = sum(a.elapsed_time(b) for a, b in tails) / iters
    ms = sum(a.elapsed_time(b) for a, b in marks) / iters - tail_ms * mlp_share
    tf = B * DIVST_GFLOP_PER_CLIP_8F * (T / 8.0) / ms          # GFLOP / ms == TFLOP/s
    return {"workload": "divided space-time attention sub-blocks of the TimeSformer forward, B=%d x %df x 224^2 (BASELINE configs[1]), %s operands" % (B, T, str(rt.compute_dtype()).replace("torch.", "")),
            "ms": round(ms, 3), "accounting": ("Block entry .. end of the spatial residual add; the fused add+norm2 kernel (%.3f ms over 12 blocks) counts %s here (its x read, delta read%s, x' write) and %s as the MLP half's norm2"
                           % (tail_ms, "6/7" if deferred else "5/6", "s" if deferred else "", "1/7" if deferred else "1/6")) if tails else "Block entry .. alpro_cls_mean_residual",
            "encoder_forward_ms": round(t0.elapsed_time(t1) / iters, 3), "gflop_per_clip": DIVST_GFLOP_PER_CLIP_8F * (T / 8.0),
            "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS["bf16"], 4), "target_frac": 0.40,
            # the conservative figure: the whole fused add + norm2 kernel counted here (no 1/6 attribution to the MLP half)
            "per_block_us": table, "measured_us_per_block": round(ms * 1e3 / 12.0, 1),
            "ms_end_to_end": round(ms + tail_ms * mlp_share, 3), "frac_end_to_end": round(B * DIVST_GFLOP_PER_CLIP_8F * (T / 8.0) / (ms + tail_ms * mlp_share) / MFMA_PEAK_TFLOPS["bf16"], 4)}


def mode_name(dtype):
    from alpro_amd import config as rt
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[dtype]
    if dtype == "fp32":
        return "fp32 exact (fp32 MFMA)"
    return "%s operands%s" % (dtype, " + precise CLS rows (fp32)" if rt.cls_precise(dt) else "")


def measure_parity(dev, dtype, full_size_backward=False):
    """Parity of the benchmarked mode, measured in THIS process against what the REFERENCE produced for the same closed-form weights and inputs
    (tests/golden/*.npz, written by tests/golden/make_golden.py from /root/reference; weights / inputs regenerated by tests/golden/det_init.py):
      * VTC logits on ALL FOUR reference fixtures that hold them (tests/golden/parity_cases.py: retrieval 2 / 16 frames -- 1 video x n captions
        through forward_inference --, pretraining 8 frames and the released 4-frame x 30-token geometry); `vtc_logits_max_abs_err` is the WORST
        of them and decides `meets_bar` (north star: within 1e-3);
      * on the retrieval fixture also ITM scores, embeddings and 451 parameter-gradient norms of itm + itc through the hand-written HIP
        backward (fp16 operands: a loss-scaled backward like the timed steps)."""
    import numpy as np
    from alpro_amd import amp, config as rt
    from tests.golden import parity_cases as pc
    from tests.test_host_cpu import make_cfg
    gdir = os.path.join(ROOT, "tests", "golden")
    err = lambda got, ref: float(np.abs(got.detach().float().cpu().numpy().astype(np.float64) - np.asarray(ref, np.float64)).max())  # noqa: E731
    orig = torch.multinomial
    torch.multinomial = lambda w, n=1, *a, **k: w.argmax(dim=-1, keepdim=True)   # the fixtures pin the hard-negative draw the same way
    prev_armed = rt._armed[0]
    res, per = {}, {}
    try:
        with rt.use_compute_dtype(dtype):
            for name in pc.CASES:
                m, batch, ref = pc.build_case(name, BERT_CFG, VENC, make_cfg, dev)
                per[name] = float("%.3e" % pc.vtc_logit_error(name, m, batch, ref))
                if name == "retrieval_T2":
                    g, gg = np.load(os.path.join(gdir, "retrieval_T2_B3.npz")), np.load(os.path.join(gdir, "retrieval_grads_T2_B3.npz"))
                    with torch.no_grad():
                        ve = m._forward_visual_embeds(batch["visual_inputs"])
                        out = m(batch)
                        inf = m.forward_inference(dict(visual_inputs=batch["visual_inputs"][:1], text_input_ids=batch["text_input_ids"],
                                                       text_input_mask=batch["text_input_mask"]))
                    res.update(itm_scores_max_abs_err=err(out["itm_scores"], g["itm_scores"]), itm_logits_inference_max_abs_err=err(inf["logits"], g["inf_logits"]),
                               itc_loss_abs_err=err(out["itc_loss"], g["itc_loss"]), video_embeds_max_abs_err=err(ve[:, [0, 1, 100, 196]], g["video_embeds_rows"]))
                    with torch.enable_grad():
                        o2 = m(batch)
                        loss = o2["itm_loss"] + o2["itc_loss"]
                        scale = 1.0
                        if amp.needs_loss_scaling():
                            sc = amp.LossScaler(init_scale=4096.0, dynamic=False, device=dev)
                            scale = 4096.0
                            with rt.loss_scaling(sc):
                                (loss * sc.scale.reshape(())).backward()
                        else:
                            loss.backward()
                    pd = dict(m.named_parameters())
                    names = [str(n) for n in gg["grad_norm_names"]]
                    got = np.array([float(pd[n].grad.norm()) / scale for n in names])
                    rel = np.abs(got - gg["grad_norms"]) / np.maximum(gg["grad_norms"], 1e-5)
                    rel[np.array([n.endswith("attention.self.key.bias") for n in names])] = 0.0   # exactly 0 in exact arithmetic (softmax shift invariance)
                    res.update(grad_norm_rel_err_worst=float(rel.max()), grad_norm_rel_err_median=float(np.median(rel)), grad_tensors=len(names),
                               grad_worst_param=names[int(rel.argmax())])
                del m, batch
                torch.cuda.empty_cache()
    finally:
        torch.multinomial = orig
        rt._armed[0] = prev_armed
    res = {k: (float("%.3e" % v) if isinstance(v, float) else v) for k, v in res.items()}
    worst = max(per, key=per.get)
    out = dict(mode=mode_name(dtype), vtc_logits_max_abs_err=per[worst], vtc_logits_worst_fixture=worst, vtc_logits_max_abs_err_per_fixture=per,
               north_star_bar="VTC logits within 1e-3 of the reference", meets_bar=bool(per[worst] <= pc.NORTH_STAR_BAR))
    out.update(res)
    # the tolerances this mode is HELD to (north_star: "within a stated fp tolerance"): the asserts of tests/test_model_parity.py for the benchmark mode
    out["stated_tolerances"] = {
        "vtc_logits_max_abs_err": pc.NORTH_STAR_BAR, "video_embeds_max_abs_err": 3e-3, "itm_scores_max_abs_err": 5e-3, "mlm_scores_max_abs_err (B=64 proxy)": 2e-2,
        "itc_loss_abs_err": 1e-3, "grad_norm_rel_err_worst (fixture, 451 tensors)": 1e-2,
        "full_size_backward (vs the exact fp32 HIP mode)": pc.FULL_SIZE_BACKWARD_LIMITS,
        "note": "fp32 exact mode: 5e-6 on every output; bf16 operands cannot meet the 1e-3 VTC bar (2.9e-3 with precise CLS rows, 9.4e-3 without: profiles/r4_parity_pareto.txt)"}
    out["within_stated_tolerances"] = bool(per[worst] <= pc.NORTH_STAR_BAR and res.get("video_embeds_max_abs_err", 0) <= 3e-3 and res.get("itm_scores_max_abs_err", 0) <= 5e-3
                                           and res.get("itc_loss_abs_err", 0) <= 1e-3 and res.get("grad_norm_rel_err_worst", 0) <= 1e-2)
    out["fixtures"] = "tests/golden/{retrieval_T2_B3, pretrain_T8_B2, retrieval_T16_B2, pretrain_release_T4_L30_B2, retrieval_grads_T2_B3}.npz (outputs of the reference itself, make_golden.py); measured in this process"
    out["full_size_proxy"] = "tests/test_model_parity.py::test_full_size_pretrain_forward_in_the_bench_dtype_vs_the_exact_mode (B=64 x 8f against the exact fp32 HIP mode); measured numbers: profiles/r4_parity_pareto.txt (forward), profiles/r5_parity_backward_B64.txt (backward)"
    if full_size_backward and dtype != "fp32":
        # the logits at the benchmarked size: max AND p99.9 of all 4096, measured here, next to meets_bar -- which stays the
        # statement about the four REFERENCE-generated fixtures; `meets_bar_at_full_size` is the same bar on the maximum at B = 64 against the exact mode
        fw = pc.full_size_forward_parity(BERT_CFG, VENC, make_cfg, dev, dtype=dtype, cls_precise=rt._cls_precise[0])
        out["full_size_forward"] = fw
        out["meets_bar_at_full_size"] = fw["max_meets_bar"]
        # the backward at the benchmarked size: every parameter gradient of one B = 64 x 8f step against the exact fp32 HIP mode (~5 s)
        bw = pc.full_size_backward_parity(BERT_CFG, VENC, make_cfg, dev, dtype=dtype, cls_precise=rt._cls_precise[0])   # (the mode the timed steps ran in)
        out["full_size_backward"] = {k: (float("%.3e" % v) if isinstance(v, float) else v) for k, v in bw.items() if k != "losses_exact"}
        out["full_size_backward"]["loss_abs_err"] = {k: float("%.2e" % v) for k, v in bw["loss_abs_err"].items()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pretrain_step", choices=["visual_fwd", "pretrain_fwd", "pretrain_step"])
    ap.add_argument("--bert-dropout", type=float, default=0.1, help="hidden/attention dropout of the BERT half in pretrain_step")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--cls-precise", default="auto", choices=["auto", "0", "1"],
                    help="precise (fp32) CLS rows of both encoders: auto = on with fp16 operands (the default mode), off with bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-divst", action="store_true", help="skip the divST sub-block measurement pass (clean per-step rocprof traces)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity measurement against tests/golden (a few seconds)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on this node (RCCL over xGMI),
        # same flags; the launched ranks find WORLD_SIZE set and fall through.  (The driver's own torchrun command line lands below directly.)
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL (see the environment notes in README.md)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, env)

    from alpro_amd import config as rt
    from alpro_amd import dist, hip
    dist.init()
    rank, world = dist.rank(), dist.size()
    assert world == args.gpus, "--gpus %d but the process group has %d rank(s): launch as `python bench.py --gpus N` or under torch.distributed.run --nproc-per-node N" % (args.gpus, world)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())  # (% only matters for gloo smoke runs)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    hip.load()
    rt.set_compute_dtype(args.dtype)
    rt.set_cls_precise(args.cls_precise)
    T = args.frames
    torch.manual_seed(1234)

    if args.workload == "visual_fwd":
        from alpro_amd.modeling.timesformer.vit import TimeSformer
        B = args.batch or 32
        model = TimeSformer(dict(VENC, num_frm=T), input_format="RGB").eval().to(dev)
        batch = synth_batch(B, T, dev, seed=rank, full=False)
        x = batch["visual_inputs"].transpose(1, 2)

        def step():
            return model.forward_features(x)
        flops_per_unit = VISUAL_GFLOP_PER_CLIP_8F * 1e9 * (T / 8.0)
        unit = "clips/s"
        wl = "TimeSformer-divST visual encoder forward (BASELINE configs[1]), B=%d x %df x 224^2" % (B, T)
    elif args.workload == "pretrain_fwd":
        from alpro_amd.modeling.alpro_models import AlproForPretrain
        B = args.batch or 64
        model = AlproForPretrain(Cfg(BERT_CFG), dict(VENC, num_frm=T)).eval().to(dev)
        batch = synth_batch(B, T, dev, seed=rank, full=True)

        def step():
            return model(batch)
        flops_per_unit = 877e9 * (T / 8.0)  # forward of one pair (3 visual passes incl. prompter, text x2, 4 fusion, heads)
        unit = "pairs/s"
        wl = "AlproForPretrain forward VTC+VTM+MLM+MPM (BASELINE configs[2], forward only; eval-mode dropout), B=%d x %df x 224^2 + 40 tok" % (B, T)
    else:
        from alpro_amd.modeling.alpro_models import AlproForPretrain
        from alpro_amd.optim import FlatAdamW
        B = args.batch or 64
        cfg = Cfg(dict(BERT_CFG, hidden_dropout_prob=args.bert_dropout, attention_probs_dropout_prob=args.bert_dropout))
        model = AlproForPretrain(cfg, dict(VENC, num_frm=T)).to(dev)
        dist.broadcast_parameters(model)
        model.train()
        opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.0, max_grad_norm=20.0)
        batch = synth_batch(B, T, dev, seed=rank, full=True)

        def step():
            out = model(batch)
            loss = out["mlm_loss"] + out["itm_loss"] + out["itc_loss"] + out["mpm_loss"]  # run_pretrain_sparse.py:557
            opt.backward(loss)  # loss.backward(); fp16 operands: the loss carries the dynamic loss scale (apex amp.scale_loss, :596-599)
            opt.step()        # gradient all-reduce (RCCL) + [1 / loss scale, overflow check] + global-norm clip + AdamW
            opt.zero_grad()
            return loss
        flops_per_unit = 1847e9 * (T / 8.0)  # 877 G forward + 970 G backward per pair
        unit = "pairs/s"
        wl = ("AlproForPretrain training step VTC+VTM+MLM+MPM fwd+bwd+allreduce+clip+AdamW (BASELINE configs[2]/[3]), "
              "B=%d pairs x %df x 224^2 + 40 tok per GPU, drop_path 0.1, BERT dropout %.2f" % (B, T, args.bert_dropout))

    train = args.workload == "pretrain_step"
    with torch.enable_grad() if train else torch.no_grad():
        for _ in range(args.warmup):
            step()
        if train:
            opt.record_exchange = True    # per-step exchange diagnostics of the timed steps (two event records per step; N > 1 only does anything)
            opt.exchange_log.clear()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
    xstats = None
    if train:
        opt.record_exchange = False
        xstats = opt.exchange_stats()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel-family device time: one extra, untimed step under HIP events.  EVERY rank runs it (the training step holds
    # collectives: a rank-0-only pass would wait forever for its peers); only rank 0 reports.
    # Launches are timed one at a time on the launch stream: the two-stream forward (alpro_amd.config.split_streams) is off for this pass, so that a
    # kernel's time is its own and not its share of the chip beside the other half's launches.
    from alpro_amd import config as _rt
    _prev_split, _prev_wgs, _prev_txt = _rt._split_streams[0], _rt.wgrad_stream_enabled(), _rt.text_stream_enabled()
    _rt.set_split_streams("0")
    _rt.set_wgrad_stream(False)   # ... and the weight gradients / the text pass stay on the launch stream for the same reason
    _rt.set_text_stream(False)
    _prev_ps = _rt.prompter_stream_enabled()
    _rt.set_prompter_stream(False)
    try:
        with (torch.enable_grad() if train else torch.no_grad()), KernelTimer(hip) as kt:
            step()
        ks = kt.summary()
    finally:
        _rt.set_split_streams(_prev_split)
        _rt.set_wgrad_stream(_prev_wgs)
        _rt.set_text_stream(_prev_txt)
        _rt.set_prompter_stream(_prev_ps)
    dist.barrier()
    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        gemm = dict(ks["gemm"])
        for fam in ("gemm_tn_acc", "gemm_qkv_tattn"):  # the wgrad GEMMs and the fused qkv + temporal-attention tiles belong to the same MFMA-bound family
            if fam in ks:
                for k_ in ("launches", "flops", "ms"):
                    gemm[k_] += ks[fam][k_]
        ach = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
        traffic, traffic_src, pm = None, None, {}
        import glob
        import re
        profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pretrain_step_B64_pmc_traffic.json")),
                       key=lambda f: (int(re.match(r"r(\d+)", os.path.basename(f)).group(1)), os.path.basename(f)))
        prof = profs[-1] if profs else ""  # the latest round's PMC passes
        if train and B == 64 and T == 8 and args.dtype in ("bf16", "fp16") and prof:
            # HBM bytes per GEMM launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
            # (tools/pmc_summary.py; FETCH_SIZE doubled per MI355X_MICROARCH.md): launch-weighted mean over the GEMM kernels
            pm = {k_.replace("alpro::", ""): v_ for k_, v_ in json.load(open(prof)).items()}
            gk = {k: v for k, v in pm.items() if k.startswith("gemm_")}
            n = sum(v["launches"] for v in gk.values())
            traffic = round(sum(v["launches"] * (v["read_bytes_corrected_per_launch"] + v["write_bytes_per_launch"]) for v in gk.values()) / n)
            traffic_src = "profiles/" + os.path.basename(prof)
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # the single dominant kernel INSTANTIATION: NT GEMM launches grouped by the template arguments the library picks --
        # persistent 256x256 kernel from 160 tiles up, activation, row map (gemm.hip launch_gemm_inst) -- with their own FLOPs, time and PMC bytes
        import re as _re
        inst = {}
        for (name, key), (c, f, ms_) in kt.shapes.items():
            if name != "gemm":
                continue
            mm = _re.match(r"M=(\d+) N=(\d+) K=(\d+) act=(\S+) res=(\d) out=(\S+) map=(\S+)", key)
            M_, N_, K_ = int(mm.group(1)), int(mm.group(2)), int(mm.group(3))
            big = ((N_ + 255) // 256) * ((M_ + 255) // 256) >= 160 and K_ >= 128
            tname = {"fp16": "f16_t", "bf16": "bf16_t", "fp32": "float"}[args.dtype]
            # round 4: identity-map 16-bit shapes with whole 256-column tiles and an even number >= 4 of K-tiles run on the 8-phase kernel --
            # 16-bit outputs with any epilogue, fp32 outputs with the plain (bias / row scale / residual) one; a ragged last tile row included
            q_ok = (args.dtype != "fp32" and mm.group(7) == "0" and N_ % 256 == 0 and K_ % 128 == 0 and K_ >= 256 and (N_ // 256) * (M_ // 256) >= 160
                    and (mm.group(6) != "float32" or mm.group(4) == "0") and hip.get_option("gemm_kind") == 1)
            if q_ok:
                kn = "gemm_nt256q_kernel<%s, %s, 0>" % (tname, mm.group(4))
            else:
                kn = "%s<%s, %s, %s%s>" % ("gemm_nt256p_kernel" if big else "gemm_nt_kernel", tname, mm.group(4), mm.group(7), ", 1" if big else "")
            d_ = inst.setdefault(kn, [0, 0.0, 0.0])
            d_[0] += c
            d_[1] += f
            d_[2] += ms_
        dom = max(inst, key=lambda k_: inst[k_][2]) if inst else None
        dominant = None
        if dom:
            c, f, ms_ = inst[dom]
            dominant = {"kernel": dom, "launches": c, "gflop_per_launch": round(f / c / 1e9, 2), "avg_launch_ms": round(ms_ / c, 4),
                        "achieved": round(f / ms_ / 1e9, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(f / ms_ / 1e9 / peak, 4),
                        "share_of_step": round(ms_ / (elapsed / args.steps * 1e3), 3), "traffic": None}
            if traffic_src and dom in pm:
                dominant["traffic"] = pm[dom]["read_bytes_corrected_per_launch"] + pm[dom]["write_bytes_per_launch"]
        shapes_tab = [{"op": n_, "shape": k_, "n": c, "ms": round(ms_, 3), "tflops": round(f / ms_ / 1e9, 1)}
                      for (n_, k_), (c, f, ms_) in sorted(kt.shapes.items(), key=lambda kv: -kv[1][2]) if f > 0][:24]
        result = {
            "metric": "video-text pairs/sec (8f x 224^2, 40-tok)", "value": round(value, 3), "unit": unit, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2), "data": "synthetic (randn clips, random token ids; random-init weights)",
            "config": {"workload": wl, "per_gpu_batch": B, "frames": T, "parallelism": ("dp%d (RCCL: feature all-gather + flat gradient all-reduce)" % world) if train else ("dp%d (independent clips, no data-path collective)" % world)},
            "mode": mode_name(args.dtype), "deterministic_reductions": bool(hip.deterministic()),
            # how the no-grad encoder forwards of the timed steps were scheduled (the visual_fwd workload itself; inside a training step the frozen
            # prompter's pass): 2 = two half batches on two HIP streams (alpro_amd.config.split_streams); the divST region is always measured on one
            "no_grad_forward_streams": 2 if _rt.split_streams(B) else 1,
            # weight-gradient GEMMs of the timed steps on a side stream beside the data-gradient chain (alpro_amd.config, ALPRO_WGRAD_STREAM)
            "wgrad_side_stream": bool(train and _rt.wgrad_stream_enabled() and (world == 1 or os.environ.get("ALPRO_WGRAD_STREAM") == "force")),   # single-GPU schedule unless forced
            "text_side_stream": bool(args.workload != "visual_fwd" and _rt.text_stream_enabled()),
            # work the reference computes and nobody reads, left out (same outputs, same gradients: tests/test_model_parity.py::test_last_fusion_layer_tail_on_the_read_rows_only;
            # ALPRO_FUSION_TAIL_ROWS=0 computes it).  model_tflops_per_gpu stays priced on the reference's FLOP count
            "dead_rows_skipped": (None if args.workload == "visual_fwd" or not getattr(model, "fusion_tail_rows", False) else
                                  "last fusion layer: attention-output dense / LayerNorms / FFN on the 239 of every 948 rows the heads read"),
            "prompter_side_stream": bool(args.workload != "visual_fwd" and _rt.prompter_stream_enabled() and (world == 1 or os.environ.get("ALPRO_PROMPTER_STREAM") == "force")),
            "world_size": world, "dist_backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else "none (single process)"),
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
            "nccl_channels": {k_: os.environ.get(k_) for k_ in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "ALPRO_OVERLAP_BACKWARD") if os.environ.get(k_) is not None},
            # the gradient exchange as rank 0 saw it over the timed steps (None at N = 1: no collective is issued).  comm_exposed_ms is the part of
            # the all-reduce NOT hidden under backward; ranges_on_wire_early / bytes_on_wire_early what went out from inside backward; cu_budget
            # the CU count the persistent grids were sized for meanwhile.  Read a SCALE record against these first.
            "exchange": xstats,
            "model_tflops_per_gpu": round(value / world * flops_per_unit / 1e12, 2),
            "roofline": {"bound": "mfma", "kernel": "gemm_nt*/gemm_tn kernels<%s> (all %d GEMM launches of one step)" % (args.dtype, gemm["launches"]),
                         "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "avg_launch_ms": round(gemm["ms"] / gemm["launches"], 4),
                         "gflop_per_launch": round(gemm["flops"] / gemm["launches"] / 1e9, 2), "traffic": traffic,
                         "traffic_unit": "HBM bytes per GEMM launch (PMC)", "traffic_source": traffic_src,
                         "dominant_instance": dominant, "gemm_shapes": shapes_tab},
            "kernel_ms_per_step": {k: round(v["ms"], 3) for k, v in ks.items()},
        }
        if world == 1 and args.dtype in ("bf16", "fp16") and T == 8 and not args.no_divst:  # the north-star kernel target, measured in the same process (~1 s)
            result["roofline"]["divst_subblock"] = measure_divst(dev, T, model if args.workload == "visual_fwd" else None)
        if train and opt.scaler is not None:
            st = opt.scaler.state.tolist()
            result["loss_scale"] = {"scale": st[0], "applied_steps": int(st[2]), "skipped_steps": int(st[3]), "policy": "dynamic (apex.amp defaults: 2^16, x0.5 on overflow, x2 per 2000 clean steps), device-resident"}
        if not args.no_parity and world == 1:  # (rank 0 alone would keep its peers waiting at N > 1)
            model = batch = None
            if train:
                opt = None
            torch.cuda.empty_cache()
            result["parity"] = measure_parity(dev, args.dtype, full_size_backward=train and B == 64 and T == 8)
        else:   # (--no-parity, or N > 1 where rank 0 alone cannot run it): say where the measured numbers of this mode live
            result["parity"] = {"mode": mode_name(args.dtype), "measured_in_this_run": False,
                                "see": "profiles/r4_parity_pareto.txt (worst-of-four-fixture VTC-logit error and the B=64 proxy per mode), profiles/r5_parity_backward_B64.txt (B=64 backward against the exact fp32 HIP mode); rerun with N=1 and without --no-parity to measure in-process"}
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (rank 0 would keep its peers waiting at N>1)
            result["cpu_baseline"] = cpu_baseline_train(T) if train else cpu_baseline(T)
        print(json.dumps(result), flush=True)
    dist.barrier()
    if world > 1 and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
