"""BERT text / fusion encoder on the MI355X kernels.

Mirrors the part of the reference's src/modeling/xbert.py that ALPRO instantiates
(BertEmbeddings :166, BertSelfAttention :216, BertSelfOutput :349, BertIntermediate :412,
BertOutput :427, BertLayer :441, BertEncoder :522 with its `mode` layer ranges :549-559,
BertModel :832 with the `encoder_embeds` bypass :1044-1053, BertLMPredictionHead :665,
BertForMaskedLM :1343): same module tree and state_dict keys, no dependency on HuggingFace
internals at run time (config is duck-typed: a transformers.BertConfig or any attribute bag).

Per layer (post-LN):  fused QKV GEMM (one (3H, H) operand built from query/key/value) ->
alpro_attn with the additive (1-mask)*-10000 key bias -> dense GEMM with the residual add fused ->
LayerNorm (fp32 + operand-dtype outputs) -> GELU GEMM -> dense GEMM + residual -> LayerNorm.
Dropout (xbert.py:212,331,358,436) is fused: hidden dropout in the dense GEMM epilogues / the embedding kernel,
attention-probability dropout inside alpro_attn; the masks are a pure hash of (seed, element index) so the backward
regenerates them instead of storing them.  Identity in eval mode.
"""
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from alpro_amd import config as rt
from alpro_amd import hip
from alpro_amd.modeling import train as tr
from alpro_amd.modeling.weights import OperandCache


def _cfg_get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.config = config


class BertSelfAttention(nn.Module):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        assert not is_cross_attention, "has_cross_attention is hard-disabled in the reference (xbert.py:450)"
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        assert self.attention_head_size == 64, "kernels are specialised for head_dim 64"
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = BertSelfAttention(config, is_cross_attention)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        assert config.hidden_act == "gelu", "only the erf GELU of config_release/base_model.json is fused"


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):
    fuse_ln_bwd_emit = os.environ.get("ALPRO_FUSE_LN_BWD", "1") != "0"   # LayerNorm backward also emits the dropped-out operand rows of the next GEMMs
    fuse_residual_ln = os.environ.get("ALPRO_FUSE_RESIDUAL_LN", "1") != "0"   # residual adds inside the post-LayerNorms (alpro_add_layernorm_fwd); False = round-2 GEMM-epilogue form (A/B)

    def __init__(self, config, layer_num):
        super().__init__()
        self.config = config
        self.attention = BertAttention(config)
        self.has_cross_attention = False
        self.layer_num = layer_num
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self._ops = OperandCache()

    def _drop(self):
        """(hidden_p, hidden_seed_attn_out, hidden_seed_ffn_out, attn_p, attn_seed): zeros in eval mode."""
        if not self.training:
            return 0.0, 0, 0, 0.0, 0
        hp, ap = float(self.config.hidden_dropout_prob), float(self.config.attention_probs_dropout_prob)
        return (hp, rt.next_dropout_seed() if hp > 0 else 0, rt.next_dropout_seed() if hp > 0 else 0,
                ap, rt.next_dropout_seed() if ap > 0 else 0)

    # ---- precise CLS rows (alpro_amd.config.cls_precise; csrc/cls_precise.hip): the [CLS] row of a text-mode layer (the row text_proj reads,
    # alpro_models.py:100-103) re-evaluated in fp32 from the layer input's [CLS] row: q | k | v (xbert.py:299-316) -> the [CLS] query's attention
    # over the caption (K / V of the other tokens as the 16-bit GEMM produced them; same key bias, same probability-dropout mask) -> dense ->
    # residual + LayerNorm (:358-359) -> intermediate GELU (:421-423) -> dense -> residual + LayerNorm (:436-437).  B fp32 rows per layer.
    # Hidden dropout: the main path's 16-bit outputs say which elements it dropped (a dropped element is exactly 0), the same elements
    # are dropped here.
    def _cls_qkv(self, hc):
        """(B, D) fp32 [CLS] rows of the layer input -> their unrounded q | k | v."""
        sa = self.attention.self
        wqkv = tr.fused_param_view([sa.query.weight, sa.key.weight, sa.value.weight])   # a view when FlatAdamW laid them out back to back
        if wqkv is None:
            wqkv = self._ops.get("qkv_w32", (sa.query.weight, sa.key.weight, sa.value.weight), torch.float32)
        # only the q third (round 6): alpro_attn_fwd's CLS query reads q from the buffer and every K / V row from the 16-bit images in LDS
        D = hc.shape[-1]
        out = torch.empty((hc.shape[0], 3 * D), dtype=torch.float32, device=hc.device)
        hip.gemm_rows(hc, wqkv[:D], bias=self._ops.get("qkv_b", (sa.query.bias, sa.key.bias, sa.value.bias), torch.float32)[:D], out=out[:, :D])
        return out

    def _cls_chain(self, hc, ctx_c, d1, d2, B, L, hp):
        """ctx_c: (B, D) fp32 attention output of the [CLS] query (alpro_attn_fwd's cls_out)."""
        f32 = torch.float32
        so = self.attention.output
        eps = self.config.layer_norm_eps
        # dense -> hidden dropout (the main path's mask, read back from its zeros) -> residual: fused into the GEMM's residual input when no
        # dropout is active, one addcmul otherwise
        def dense_res(a, w, bias, res, d_main):
            if hp <= 0:
                return hip.gemm_rows(a, w, bias=bias, residual=res)
            keep = (d_main.view(B, L, -1)[:, 0] != 0).to(f32).mul_(1.0 / (1.0 - hp))
            return torch.addcmul(res, hip.gemm_rows(a, w, bias=bias), keep)
        s1_c = dense_res(ctx_c, self._ops.get("ao_w", so.dense.weight, f32), so.dense.bias, hc, d1)
        a32_c = hip.layernorm(s1_c, so.LayerNorm.weight, so.LayerNorm.bias, eps, f32)
        it_c = hip.gemm_rows(a32_c, self._ops.get("i_w", self.intermediate.dense.weight, f32), bias=self.intermediate.dense.bias, act=hip.ACT_GELU)
        s2_c = dense_res(it_c, self._ops.get("o_w", self.output.dense.weight, f32), self.output.dense.bias, a32_c, d2)
        o32_c = hip.layernorm(s2_c, self.output.LayerNorm.weight, self.output.LayerNorm.bias, eps, f32)
        return s1_c, a32_c, s2_c, o32_c

    def forward(self, h32, h_t, key_bias, B, L, rows=None):
        """h32 (B*L, H) fp32 residual stream, h_t the same in the operand dtype; returns the next pair (of the `rows` only, if given)."""
        o32, o_t, _ = self.forward_train(h32, h_t, key_bias, B, L, save=False, rows=rows)
        return o32, o_t

    # ---- training path ---------------------------------------------------------------------------------
    def forward_train(self, h32, h_t, key_bias, B, L, save=True, rows=None):
        """rows (optional, LongTensor of flat row indices): the only rows of this layer's OUTPUT anybody reads (the last fusion layer: [CLS] rows, the
        MLM pairs' text rows, the positives' patch rows -- a quarter of the 4B x 237).  Attention still runs over every row (all keys / values are
        needed, and the kernel's unit is a whole sequence); everything behind it -- attention-output dense, both LayerNorms, the FFN -- is row-wise
        and runs on the gathered rows only; the outputs are then (len(rows), D)."""
        hp, seed1, seed2, ap, seed_a = self._drop()
        dt = rt.compute_dtype()
        sa, so = self.attention.self, self.attention.output
        eps = self.config.layer_norm_eps
        H = sa.num_attention_heads
        scale = 1.0 / math.sqrt(sa.attention_head_size)
        wqkv = self._ops.get("qkv_w", (sa.query.weight, sa.key.weight, sa.value.weight), dt)
        bqkv = self._ops.get("qkv_b", (sa.query.bias, sa.key.bias, sa.value.bias), torch.float32)
        qkv = hip.gemm(h_t, wqkv, bias=bqkv)
        cp = rt.cls_precise(dt) and self.fuse_residual_ln and self.layer_num < int(_cfg_get(self.config, "fusion_layer", 0) or 0)
        if cp:   # the [CLS] query once more in fp32, inside the same attention launch
            hc = h32.view(B, L, -1)[:, 0]
            ctx, lse, ctx_c = hip.attn(qkv, B, L, H, scale, key_bias, want_lse=True, drop_p=ap, drop_seed=seed_a, cls_q=self._cls_qkv(hc), cls_group=1)
        else:
            ctx, lse = hip.attn(qkv, B, L, H, scale, key_bias, want_lse=True, drop_p=ap, drop_seed=seed_a)
        ctx_full, M_full = ctx, h_t.shape[0]
        if rows is not None and (cp or not self.fuse_residual_ln):
            raise RuntimeError("BertLayer: an output row subset is supported on the fused residual + LayerNorm path of the fusion layers only")
        if rows is not None:
            ctx = ctx.index_select(0, rows)
            h32 = h32.index_select(0, rows)
        u, u_tiled = tr.gelu_save_buffer(ctx.shape[0], self.intermediate.dense.out_features, h_t.shape[1], dt, h_t.device) if save else (None, False)
        if self.fuse_residual_ln:
            # the two dense Linears write their (dropped-out) 16-bit output only; residual add + post-LayerNorm are one streaming kernel
            # (alpro_add_layernorm_fwd), which also leaves the pre-LayerNorm sums s1 / s2 the backward needs (training only)
            d1 = hip.gemm(ctx, self._ops.get("ao_w", so.dense.weight, dt), bias=so.dense.bias, drop_p=hp, drop_seed=seed1)
            a_t, a32, s1 = hip.add_layernorm(h32, d1, so.LayerNorm.weight, so.LayerNorm.bias, eps, out32=True, want_x=save)
            it = hip.gemm(a_t, self._ops.get("i_w", self.intermediate.dense.weight, dt), bias=self.intermediate.dense.bias, act=(hip.ACT_GELU_SAVE_GRAD if (save and tr.SAVE_GELU_GRAD) else hip.ACT_GELU), pre_act=u, c2_tiled=u_tiled)
            d2 = hip.gemm(it, self._ops.get("o_w", self.output.dense.weight, dt), bias=self.output.dense.bias, drop_p=hp, drop_seed=seed2)
            o_t, o32, s2 = hip.add_layernorm(a32, d2, self.output.LayerNorm.weight, self.output.LayerNorm.bias, eps, out32=True, want_x=save)
            if cp:
                D = h32.shape[1]
                s1_c, a32_c, s2_c, o32_c = self._cls_chain(hc, ctx_c, d1, d2, B, L, hp if seed1 else 0.0)
                o32.view(B, L, D)[:, 0] = o32_c
                o_t.view(B, L, D)[:, 0].copy_(o32_c)          # (copy_ converts: one launch instead of .to() + copy)
                a_t.view(B, L, D)[:, 0].copy_(a32_c)          # (the FFN's saved input row, for its weight gradient)
                if save:                                     # the two LayerNorm backward inputs
                    s1.view(B, L, D)[:, 0] = s1_c
                    s2.view(B, L, D)[:, 0] = s2_c
        else:
            s1 = hip.gemm(ctx, self._ops.get("ao_w", so.dense.weight, dt), bias=so.dense.bias, out_dtype=torch.float32, residual=h32,
                          drop_p=hp, drop_seed=seed1)
            a_t, a32 = hip.layernorm(s1, so.LayerNorm.weight, so.LayerNorm.bias, eps, dt, out32=True)
            it = hip.gemm(a_t, self._ops.get("i_w", self.intermediate.dense.weight, dt), bias=self.intermediate.dense.bias, act=(hip.ACT_GELU_SAVE_GRAD if (save and tr.SAVE_GELU_GRAD) else hip.ACT_GELU), pre_act=u, c2_tiled=u_tiled)
            s2 = hip.gemm(it, self._ops.get("o_w", self.output.dense.weight, dt), bias=self.output.dense.bias, out_dtype=torch.float32, residual=a32,
                          drop_p=hp, drop_seed=seed2)
            o_t, o32 = hip.layernorm(s2, self.output.LayerNorm.weight, self.output.LayerNorm.bias, eps, dt, out32=True)
        sv = dict(h_t=h_t, qkv=qkv, ctx=ctx, lse=lse, s1=s1, a_t=a_t, u=u, u_grad=tr.SAVE_GELU_GRAD, u_tiled=u_tiled, it=it, s2=s2, kb=key_bias, dims=(B, L, H, scale), dt=dt,
                  drop=(hp, seed1, seed2, ap, seed_a), rows=rows, ctx_full=ctx_full, M_full=M_full) if save else None
        return o32, o_t, sv

    def backward(self, sv, do32, do_t):
        """Gradients w.r.t. the layer's two output streams (fp32 residual copy, operand-dtype copy; either may be
        None) -> (dh32, dh_t) for the layer input.  Parameter gradients accumulate into .grad."""
        B, L, H, scale = sv["dims"]
        dt = sv["dt"]
        sa, so = self.attention.self, self.attention.output
        eps = self.config.layer_norm_eps
        dev = sv["s1"].device
        M, D = sv["s1"].shape

        def ln_bwd(ln, x, dy_t, dy32, drop_seed):
            """-> (dx fp32, dx through the dropout of the dense output that feeds this LayerNorm, in the operand dtype)"""
            if dy_t is None:
                dy_t, dy32 = dy32, None
            dx = torch.empty((M, D), dtype=torch.float32, device=dev)
            g, b_ = tr.grad_buffer(ln.weight, zero=True)[0], tr.grad_buffer(ln.bias, zero=True)[0]
            if self.fuse_ln_bwd_emit:
                _, dx_t = hip.layernorm_bwd(dy_t, x, ln.weight, eps, dx, g, b_, dy2=dy32, accumulate=False,
                                            emit=dict(mode=hip.EMIT_ROWS, rows=M, dtype=dt, drop_p=hp if drop_seed else 0.0, drop_seed=drop_seed))
                return dx, dx_t
            hip.layernorm_bwd(dy_t, x, ln.weight, eps, dx, g, b_, dy2=dy32, accumulate=False)
            return dx, hip.gather_cast(dx, dt, drop_p=hp, drop_seed=drop_seed)

        hp, seed1, seed2, ap, seed_a = sv["drop"]
        ds2, ds2_t = ln_bwd(self.output.LayerNorm, sv["s2"], do_t, do32, seed2)   # ds2 is also d(a32): identity residual; ds2_t: through the FFN-output dropout
        tr.wgrad(ds2_t, sv["it"], self.output.dense.weight, self.output.dense.bias)
        du = tr.dgrad(ds2_t, tr.transposed_operand(self._ops, "o_w^T", self.output.dense.weight, dt), gelu_pre=sv["u"], gelu_saved_grad=sv.get("u_grad", False), gelu_tiled=sv.get("u_tiled", False))
        tr.wgrad(du, sv["a_t"], self.intermediate.dense.weight, self.intermediate.dense.bias)
        da_t = tr.dgrad(du, tr.transposed_operand(self._ops, "i_w^T", self.intermediate.dense.weight, dt))
        ds1, ds1_t = ln_bwd(so.LayerNorm, sv["s1"], da_t, ds2, seed1)       # ds1 is also d(h32): identity residual; ds1_t: through the attention-output dropout
        tr.wgrad(ds1_t, sv["ctx"], so.dense.weight, so.dense.bias)
        dctx = tr.dgrad(ds1_t, tr.transposed_operand(self._ops, "ao_w^T", so.dense.weight, dt))
        rows = sv.get("rows")
        if rows is not None:   # the row-wise tail ran on the gathered output rows only: their gradients go back to their places, every other row's is zero
            full = torch.zeros((sv["M_full"], D), dtype=dctx.dtype, device=dev)
            full.index_copy_(0, rows, dctx)
            dctx = full
            full32 = torch.zeros((sv["M_full"], D), dtype=torch.float32, device=dev)
            full32.index_copy_(0, rows, ds1)
            ds1 = full32
        dqkv = hip.attn_bwd(sv["qkv"], sv["ctx_full"], dctx, sv["lse"], B, L, H, scale, sv["kb"], drop_p=ap, drop_seed=seed_a)
        # fused q/k/v projection: the three weight gradients come from the three column blocks of dqkv
        Hd = sa.all_head_size
        lins = (sa.query, sa.key, sa.value)
        gw = gb = None
        if dqkv.dtype != torch.float32:
            for lin in lins:
                tr.grad_buffer(lin.weight, zero=True)
                tr.grad_buffer(lin.bias, zero=True)
            gw, gb = tr.fused_grad_view([l.weight for l in lins]), tr.fused_grad_view([l.bias for l in lins])
        if gw is not None and gb is not None:  # flat gradient buffer: one (3H, H) weight-gradient GEMM + bias gradient
            hip.gemm_tn_acc(dqkv, sv["h_t"], gw, colsum=gb)
        else:
            for i, lin in enumerate(lins):
                tr.wgrad(dqkv[:, i * Hd:(i + 1) * Hd], sv["h_t"], lin.weight, lin.bias)
        wT = (None, tr.transposed_operand(self._ops, "qkv_w^T", (sa.query.weight, sa.key.weight, sa.value.weight), dt))
        dh_t = tr.dgrad(dqkv, wT[1])
        return ds1, dh_t


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config, i) for i in range(config.num_hidden_layers)])

    def layer_range(self, mode):
        if mode == 'text':
            return 0, self.config.fusion_layer
        if mode == 'fusion':
            return self.config.fusion_layer, self.config.num_hidden_layers
        return 0, self.config.num_hidden_layers  # 'multi_modal' (xbert.py:557-559)


def _resolve_pretrained(name_or_path):
    """Local file holding the weights `name_or_path` stands for, or None (see BertPreTrainedModel.from_pretrained)."""
    import os
    names = ("pytorch_model.bin", "model.safetensors")

    def pick(p):
        if not isinstance(p, str) or not p:
            return None
        if os.path.isfile(p):
            return p
        if os.path.isdir(p):
            for n in names:
                if os.path.isfile(os.path.join(p, n)):
                    return os.path.join(p, n)
        return None

    cands = [name_or_path, os.environ.get("ALPRO_BERT_WEIGHTS")]
    if os.environ.get("ALPRO_PRETRAINED_DIR") and isinstance(name_or_path, str):
        cands.append(os.path.join(os.environ["ALPRO_PRETRAINED_DIR"], name_or_path))
    for c in cands:
        hit = pick(c)
        if hit:
            return hit
    try:
        from huggingface_hub import try_to_load_from_cache
        for n in names:
            hit = try_to_load_from_cache(name_or_path, n)
            if isinstance(hit, str) and os.path.isfile(hit):
                return hit
    except Exception:  # no hub package / malformed repo id: the cache is simply not a source
        pass
    return None


class BertPreTrainedModel(nn.Module):
    """Weight init of xbert.py:728-738 (normal(0, initializer_range), LN ones/zeros, zero biases)."""

    def __init__(self, config):
        super().__init__()
        self.config = config

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def init_weights(self):
        self.apply(self._init_weights)

    @classmethod
    def from_pretrained(cls, name_or_path, config=None, **kwargs):
        """Reference call sites (alpro_models.py:30,637) pass 'bert-base-uncased' and get the HF weights.  There is no network
        on the GPU boxes, so the name is resolved LOCALLY, in this order: (1) `name_or_path` itself if it is a file / directory,
        (2) $ALPRO_BERT_WEIGHTS (file or directory), (3) $ALPRO_PRETRAINED_DIR/<name>/, (4) the Hugging Face cache
        (huggingface_hub.try_to_load_from_cache).  Files: `pytorch_model.bin`, `model.safetensors`, or a torch-saved state_dict.
        If nothing is found the model keeps the random init of xbert.py:728-738 and says so LOUDLY (warning; RuntimeError when
        ALPRO_REQUIRE_PRETRAINED=1) -- a pretrain / finetune run that silently starts from random BERT weights does not reproduce
        the reference.  Missing and unexpected keys are reported, never swallowed."""
        import logging
        import os
        import warnings
        log = logging.getLogger(__name__)
        model = cls(config, **kwargs)
        path = _resolve_pretrained(name_or_path)
        if path is None:
            msg = ("%s.from_pretrained(%r): no local weights found (looked at the path itself, $ALPRO_BERT_WEIGHTS, $ALPRO_PRETRAINED_DIR and "
                   "the Hugging Face cache); the text encoder starts from RANDOM initialisation" % (cls.__name__, name_or_path))
            if os.environ.get("ALPRO_REQUIRE_PRETRAINED", "0") == "1":
                raise RuntimeError(msg)
            warnings.warn(msg)
            log.warning(msg)
            return model
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            sd = torch.load(path, map_location="cpu")
        sd = {k.replace("gamma", "weight").replace("beta", "bias"): v for k, v in sd.items()}
        own = model.state_dict()
        if not any(k in own for k in sd) and any(k.startswith("bert.") for k in sd):
            # a BertForMaskedLM-style checkpoint ('bert.encoder...') loaded into the bare BertModel (alpro_models.py:637)
            sd = {k[len("bert."):]: v for k, v in sd.items() if k.startswith("bert.")}
        res = model.load_state_dict(sd, strict=False)
        model.tie_weights()
        model.pretrained_report = dict(path=path, missing=list(res.missing_keys), unexpected=list(res.unexpected_keys))
        loaded = len(own) - len(res.missing_keys)
        log.info("%s.from_pretrained: %s -> %d / %d tensors loaded; missing %d %s; unexpected %d %s", cls.__name__, path, loaded, len(own),
                 len(res.missing_keys), res.missing_keys[:6], len(res.unexpected_keys), res.unexpected_keys[:6])
        if loaded == 0:
            raise RuntimeError("%s.from_pretrained: %s shares no key with the model (first keys: %s)" % (cls.__name__, path, list(sd)[:4]))
        return model

    def tie_weights(self):
        pass


class BertModel(BertPreTrainedModel):
    def __init__(self, config, add_pooling_layer=True):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = None  # ALPRO builds BertForMaskedLM(add_pooling_layer=False) (xbert.py:1352)
        self.init_weights()

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    @staticmethod
    def key_bias(attention_mask):
        """(B, L) {0,1} -> additive fp32 bias (1 - m) * -10000 (xbert.py:936-937)."""
        return ((1.0 - attention_mask.to(torch.float32)) * -10000.0).contiguous()

    def forward(self, input_ids=None, attention_mask=None, encoder_embeds=None, return_dict=True, mode='multi_modal', encoder_embeds_parts=None, out_rows=None, **unused):
        """encoder_embeds_parts (this repo's extension; the reference passes encoder_embeds): (text_pool (Pt, Lt, D), video_pool (Pv, Lv, D), ti (S,),
        vi (S,)) -- the fusion batch as a gather, sequence s = [text_pool[ti[s]] ; video_pool[vi[s]]], i.e. what the reference's
        torch.cat([text_embeds, video_embeds], dim=1) over concatenated / index-selected batches holds (alpro_models.py:278-281,325-330,360-363),
        built by one kernel and differentiated by one (alpro_gather_seq_fwd / _bwd) instead of materialised by torch.cat and autograd.
        out_rows (this repo's extension): flat indices into the (B * L) output rows -- last_hidden_state is then (len(out_rows), D), those rows only, and
        the last layer's row-wise tail runs on them alone (BertLayer.forward_train)."""
        parts = encoder_embeds_parts
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            lo, hi = self.encoder.layer_range(mode)
            params = [p for i in range(lo, hi) for p in self.encoder.layer[i].parameters()]
            if encoder_embeds is None and parts is None:
                params += list(self.embeddings.parameters())
            run = _BertRun(self, input_ids, attention_mask, mode)
            run.out_rows = out_rows
            if parts is not None:
                run.parts = (parts[2].contiguous(), parts[3].contiguous())
                out = tr.run_anchored(run, [parts[0], parts[1]], params)
            else:
                out = tr.run_anchored(run, [encoder_embeds] if encoder_embeds is not None else [], params)
            if not return_dict:
                return (out,)
            return SimpleNamespace(last_hidden_state=out, pooler_output=None, hidden_states=None, attentions=None,
                                   past_key_values=None, cross_attentions=None)
        dt = rt.compute_dtype()
        cfg = self.config
        if parts is not None:
            B, L = parts[2].numel(), parts[0].shape[1] + parts[1].shape[1]
            h32, h_t = hip.gather_seq(parts[0].contiguous().float(), parts[1].contiguous().float(), parts[2].contiguous(), parts[3].contiguous(), dt)
            if h_t is None:
                h_t = h32
        elif encoder_embeds is None:
            B, L = input_ids.shape
            emb = self.embeddings
            ep = float(cfg.hidden_dropout_prob) if emb.training else 0.0
            h32, h_t = hip.bert_embed(input_ids.contiguous(), emb.word_embeddings.weight, emb.position_embeddings.weight,
                                      emb.token_type_embeddings.weight, emb.LayerNorm.weight, emb.LayerNorm.bias, cfg.layer_norm_eps, dt,
                                      drop_p=ep, drop_seed=rt.next_dropout_seed() if ep > 0 else 0)
        else:
            B, L, Hd = encoder_embeds.shape
            h32 = encoder_embeds.reshape(B * L, Hd).contiguous().float()
            h_t = hip.cast(h32, dt)
        if attention_mask is None:
            attention_mask = torch.ones((B, L), device=h32.device)
        kb = self.key_bias(attention_mask)
        lo, hi = self.encoder.layer_range(mode)
        for i in range(lo, hi):
            h32, h_t = self.encoder.layer[i](h32, h_t, kb, B, L, rows=out_rows if i == hi - 1 else None)
        out = h32.view(B, L, -1) if out_rows is None else h32
        if not return_dict:
            return (out,)
        return SimpleNamespace(last_hidden_state=out, pooler_output=None, hidden_states=None, attentions=None,
                               past_key_values=None, cross_attentions=None)


class _BertRun:
    """Forward/backward of one text-mode or fusion-mode pass of BertModel for tr.Anchor."""

    def __init__(self, model, input_ids, attention_mask, mode):
        self.m, self.ids, self.mask, self.mode = model, input_ids, attention_mask, mode
        self.parts = None   # (ti, vi): the input is a gather of (text pool, video pool) sequences, the two activations of forward()
        self.out_rows = None   # flat indices of the only output rows the caller reads (BertModel.forward out_rows)

    def forward(self, encoder_embeds=None, video_pool=None):
        m, cfg = self.m, self.m.config
        dt = rt.compute_dtype()
        emb = m.embeddings
        if self.parts is not None:
            text_pool = encoder_embeds
            ti, vi = self.parts
            B, L = ti.numel(), text_pool.shape[1] + video_pool.shape[1]
            self.pool_dims = (text_pool.shape[0], video_pool.shape[0], text_pool.shape[1], video_pool.shape[1])
            h32, h_t = hip.gather_seq(text_pool.contiguous().float(), video_pool.contiguous().float(), ti, vi, dt)
            if h_t is None:
                h_t = h32
        elif encoder_embeds is None:
            B, L = self.ids.shape
            self.ids = self.ids.contiguous()
            word = emb.word_embeddings.weight
            ep = float(cfg.hidden_dropout_prob) if emb.training else 0.0
            self.emb_drop = (ep, rt.next_dropout_seed() if ep > 0 else 0)
            # pre-LayerNorm sum is needed by the LN backward: recompute it there from the tables (cheap gather)
            h32, h_t = hip.bert_embed(self.ids, word, emb.position_embeddings.weight, emb.token_type_embeddings.weight,
                                      emb.LayerNorm.weight, emb.LayerNorm.bias, cfg.layer_norm_eps, dt,
                                      drop_p=self.emb_drop[0], drop_seed=self.emb_drop[1])
        else:
            B, L, Hd = encoder_embeds.shape
            h32 = encoder_embeds.reshape(B * L, Hd).contiguous().float()
            h_t = hip.cast(h32, dt)
        mask = self.mask if self.mask is not None else torch.ones((B, L), device=h32.device)
        kb = m.key_bias(mask)
        self.dims = (B, L)
        self.from_ids = encoder_embeds is None and self.parts is None
        self.saved = []
        lo, hi = m.encoder.layer_range(self.mode)
        self.range = (lo, hi)
        for i in range(lo, hi):
            h32, h_t, sv = m.encoder.layer[i].forward_train(h32, h_t, kb, B, L, rows=self.out_rows if i == hi - 1 else None)
            self.saved.append(sv)
        return h32.view(B, L, -1) if self.out_rows is None else h32

    def backward(self, dout):
        m, cfg = self.m, self.m.config
        B, L = self.dims
        lo, hi = self.range
        d32, d_t = (dout.reshape(B * L, -1) if self.out_rows is None else dout.reshape(self.out_rows.numel(), -1)).contiguous(), None
        for i in range(hi - 1, lo - 1, -1):
            d32, d_t = m.encoder.layer[i].backward(self.saved.pop(), d32, d_t)
        if self.parts is not None:   # per pool row, the sum over the sequences that used it (and of the 16-bit + fp32 parts of the gradient)
            if d_t is not None and d_t.dtype == torch.float32:
                d32, d_t = d32 + d_t, None
            return hip.gather_seq_bwd(d32.contiguous(), None if d_t is None else d_t.contiguous(), self.parts[0], self.parts[1], *self.pool_dims)
        if not self.from_ids:
            return (d32 + d_t.float()).view(B, L, -1)
        # embeddings: LN backward on word + type0 + pos, then scatter the row gradients into the tables
        emb = m.embeddings
        D = d32.shape[1]
        pre = (emb.word_embeddings.weight.detach()[self.ids.view(-1)] + emb.token_type_embeddings.weight.detach()[0]
               + emb.position_embeddings.weight.detach()[:L].repeat(B, 1)).contiguous()
        de = torch.empty_like(pre)
        g, b_ = tr.grad_buffer(emb.LayerNorm.weight, zero=True)[0], tr.grad_buffer(emb.LayerNorm.bias, zero=True)[0]
        hip.layernorm_bwd(d_t, pre, emb.LayerNorm.weight, cfg.layer_norm_eps, de, g, b_, dy2=d32, accumulate=False,
                          drop_p=self.emb_drop[0], drop_seed=self.emb_drop[1])
        gw = tr.grad_buffer(emb.word_embeddings.weight, zero=True)[0] if emb.word_embeddings.weight.grad is None else emb.word_embeddings.weight.grad
        pad = emb.word_embeddings.padding_idx   # nn.Embedding(padding_idx): the pad row's lookup gradient stays zero (xbert.py:171)
        hip.scatter_add_rows(de, self.ids.view(-1), gw, skip_idx=-1 if pad is None else int(pad))
        gp = tr.grad_buffer(emb.position_embeddings.weight, zero=True)[0] if emb.position_embeddings.weight.grad is None else emb.position_embeddings.weight.grad
        hip.scatter_add_rows(de, None, gp, idx_mod=L)
        gt = tr.grad_buffer(emb.token_type_embeddings.weight, zero=True)[0] if emb.token_type_embeddings.weight.grad is None else emb.token_type_embeddings.weight.grad
        gt[0].add_(de.sum(0))
        return None


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias  # xbert.py:677
        self._ops = OperandCache()

    def forward(self, hidden_states):
        """(B, Lt, H) fp32 -> (B, Lt, vocab) fp32 logits (xbert.py:679-682)."""
        if torch.is_grad_enabled() and (hidden_states.requires_grad or any(p.requires_grad for p in self.parameters())):
            return tr.run_anchored(_LMHeadRun(self), [hidden_states], list(self.parameters()))
        dt = rt.compute_dtype()
        shp = hidden_states.shape
        h = hip.cast(hidden_states.reshape(-1, shp[-1]).contiguous().float(), dt)
        t = self.transform
        g = hip.gemm(h, self._ops.get("t_w", t.dense.weight, dt), bias=t.dense.bias, act=hip.ACT_GELU, out_dtype=torch.float32)
        n = hip.layernorm(g, t.LayerNorm.weight, t.LayerNorm.bias, self.config.layer_norm_eps, dt)
        logits = hip.gemm(n, self._ops.get("dec_w", self.decoder.weight, dt), bias=self.bias, out_dtype=torch.float32)
        return logits.view(*shp[:-1], -1)

    def forward_with_loss(self, hidden_states, labels, ignore_index=-100):
        """(logits, mean cross-entropy over labels != ignore_index): BertLMPredictionHead + CrossEntropyLoss of
        alpro_models.py:368-371 as ONE autograd node -- alpro_softmax_xent writes (softmax - onehot)/n straight in the
        operand dtype, so the (B*Lt, vocab) fp32 gradient is never materialised.  NaN when no label is valid, like the
        reference."""
        run = _LMHeadRun(self, labels.reshape(-1).contiguous(), ignore_index)
        need = torch.is_grad_enabled() and (hidden_states.requires_grad or any(p.requires_grad for p in self.parameters()))
        if need:
            logits, loss = tr.run_anchored(run, [hidden_states], list(self.parameters()))
        else:
            with torch.no_grad():
                logits, loss = run.forward(hidden_states)
        return logits, loss


class _LMHeadRun:
    """Forward/backward of BertLMPredictionHead for tr.Anchor; the decoder weight is the word-embedding table."""

    def __init__(self, head, labels=None, ignore_index=-100):
        self.h, self.labels, self.ignore = head, labels, ignore_index

    def forward(self, hidden):
        hd, dt = self.h, rt.compute_dtype()
        self.shape = hidden.shape
        t = hd.transform
        x = hip.cast(hidden.reshape(-1, self.shape[-1]).contiguous().float(), dt)
        u = torch.empty((x.shape[0], t.dense.out_features), dtype=dt, device=x.device)
        g = hip.gemm(x, hd._ops.get("t_w", t.dense.weight, dt), bias=t.dense.bias, act=hip.ACT_GELU, out_dtype=torch.float32, pre_act=u)
        n = hip.layernorm(g, t.LayerNorm.weight, t.LayerNorm.bias, hd.config.layer_norm_eps, dt)
        logits = hip.gemm(n, hd._ops.get("dec_w", hd.decoder.weight, dt), bias=hd.bias, out_dtype=torch.float32)
        self.x, self.u, self.g, self.n, self.dt = x, u, g, n, dt
        if self.labels is None:
            return logits.view(*self.shape[:-1], -1)
        inv_n = (1.0 / (self.labels != self.ignore).sum().to(torch.float32)).reshape(1)
        want_grad = any(p.requires_grad for p in hd.parameters())
        # fp16 operands: (softmax - onehot) / n is written in 16 bits NOW, so it must already carry the loss scale of the coming
        # backward (alpro_amd.amp); backward() divides the incoming d(loss) by the same value.
        self.pre_scale = rt.armed_loss_scale(logits.device) if want_grad else None
        if want_grad:
            loss_rows, self.dl = hip.softmax_xent(logits, self.labels, grad_dtype=dt, grad_scale=inv_n if self.pre_scale is None else inv_n * self.pre_scale,
                                                  ignore_index=self.ignore)
        else:
            loss_rows, self.dl = hip.softmax_xent(logits, self.labels, ignore_index=self.ignore), None
        return logits.view(*self.shape[:-1], -1), (loss_rows.sum() * inv_n).reshape(())

    def backward(self, dlogits, dloss=None):
        hd, dt = self.h, self.dt
        t = hd.transform
        M, V = self.x.shape[0], hd.decoder.weight.shape[0]
        Vp = (V + 63) // 64 * 64
        up = None
        if self.labels is not None:
            dl = self.dl                       # (softmax - onehot) / n_valid, already in the operand dtype
            if dloss is not None:
                up = dloss.reshape(()).float()        # upstream scale of the loss (1 in the reference's sum of losses; the loss scale under fp16)
                if getattr(self, "pre_scale", None) is not None:
                    up = up / self.pre_scale.reshape(())   # already applied at forward time
            if up is not None and dt == torch.float16 and getattr(self, "pre_scale", None) is None:
                # fp16 operands whose loss scale was NOT known at forward time (a scaler armed after the forward): `up` then IS the loss scale
                # (2^16) and would overflow the 16-bit (M, H) tensors below -- apply it to the small probabilities instead
                dl = dl * up
                up = None
            if dlogits is not None:            # someone also differentiated through mlm_scores
                dl = (dl * up) if up is not None else dl.clone()
                up = None
                dl[:, :V] += dlogits.reshape(M, V).to(dt)
            # otherwise `up` (a device scalar, == 1 in the usual sum of losses) is NOT multiplied into the (M, vocab) tensor -- a 0.3 GB pass for a
            # factor of one -- but into the three small things that are linear in dl: the decoder dgrad's (M, H) result, the (M, H) operand of
            # the tied weight's gradient, and the bias column sums
        else:
            dl = torch.zeros((M, Vp), dtype=dt, device=dlogits.device)
            dl[:, :V] = dlogits.reshape(M, V)
        dn = tr.dgrad(dl, tr.transposed_operand(hd._ops, "dec_w^T", hd.decoder.weight, dt))
        n_op = self.n
        if up is not None:
            dn = dn * up
            n_op = n_op * up
        # decoder weight is the (tied) word-embedding table: dW (V, H) += dl^T n ; bias += colsum(dl)
        gw = tr.grad_buffer(hd.decoder.weight, zero=True)[0]
        cs = torch.zeros(Vp, dtype=torch.float32, device=dl.device)
        if dt != torch.float32:
            hip.gemm_tn_acc(dl[:, :V], n_op, gw, colsum=cs)
        else:
            dlT = hip.transpose(dl, colsum=cs)
            hip.gemm(dlT[:V], hip.transpose(n_op), out=gw, out_dtype=torch.float32, residual=gw)
        tr.add_grad(hd.bias, cs[:V] if up is None else cs[:V] * up)
        dg = torch.empty_like(self.g)
        gw, gb = tr.grad_buffer(t.LayerNorm.weight, zero=True)[0], tr.grad_buffer(t.LayerNorm.bias, zero=True)[0]
        hip.layernorm_bwd(dn, self.g, t.LayerNorm.weight, hd.config.layer_norm_eps, dg, gw, gb, accumulate=False)
        du = hip.gelu_bwd(hip.gather_cast(dg, dt), self.u)
        tr.wgrad(du, self.x, t.dense.weight, t.dense.bias)
        dx = tr.dgrad(du, tr.transposed_operand(hd._ops, "t_w^T", t.dense.weight, dt), out_dtype=torch.float32)
        self.x = self.u = self.g = self.n = None
        return dx.view(self.shape)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertForMaskedLM(BertPreTrainedModel):
    def __init__(self, config):
        super().__init__(config)
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(config)
        self.init_weights()
        self.tie_weights()

    def tie_weights(self):
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight  # xbert.py:670-677 + HF tie

    def get_input_embeddings(self):
        return self.bert.embeddings.word_embeddings

    def get_output_embeddings(self):
        return self.cls.predictions.decoder
