"""CPU oracle: a plain-PyTorch fp32 restatement of ALPRO's video-text hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the *checker*; nothing under alpro_amd/ imports it and the product path
never falls back to it.

Parity pin: the reference ships no tests or golden vectors, so this
oracle is pinned against outputs of the reference itself, captured by importing
/root/reference on CPU in the authoring container (tests/golden/make_golden.py ->
tests/golden/*.npz, checked by tests/test_oracle_golden.py).

Every function is functional over a flat `p` dict that uses the reference's state_dict key
names, and cites the reference lines it restates (paths relative to
/root/reference).  Stochastic pieces (dropout, drop_path, multinomial negatives) are exposed
as explicit arguments so that parity runs are deterministic: eval semantics by default.
"""
import math

import torch
import torch.nn.functional as F

NEG_INF = float("-inf")


# ----------------------------------------------------------------------------- primitives
def linear(x, p, name):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def layer_norm(x, p, name, eps):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], eps)


def gelu_erf(x):
    # transformers ACT2FN['gelu'] == exact erf GELU; nn.GELU() default likewise (vit.py:50, xbert.py:417)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# ----------------------------------------------------------------------------- TimeSformer
VIT_EPS = 1e-6  # vit.py:453
VIT_HEADS = 12  # vit.py:450


def vit_attention(x, p, name, num_heads=VIT_HEADS):
    """src/modeling/timesformer/vit.py:81-100 (Attention.forward, with_qkv=True, attn_drop=0)."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = linear(x, p, name + ".qkv").reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return linear(x, p, name + ".proj")


def _row_scale(x, keep):
    """drop_path (vit_utils.py:137-151) with an explicit per-row multiplier (None == eval)."""
    if keep is None:
        return x
    return x * keep.reshape((-1,) + (1,) * (x.ndim - 1))


def vit_block(x, p, name, B, T, W, drop=None):
    """vit.py:136-213 (Block.forward, divided_space_time).

    x: (B, 1 + H*W*T, D) with patch token (h, w, t) at index 1 + (h*W + w)*T + t.
    drop: optional dict of drop_path multipliers {'t': (B*H*W,), 's': (B*T,), 'm': (B,)}.
    """
    drop = drop or {}
    n_sp = (x.size(1) - 1) // T
    D = x.size(2)
    # temporal attention over the T frames of each patch (vit.py:146-162)
    xt = x[:, 1:, :].reshape(B * n_sp, T, D)
    res_t = vit_attention(layer_norm(xt, p, name + ".temporal_norm1", VIT_EPS), p, name + ".temporal_attn")
    res_t = _row_scale(res_t, drop.get("t")).reshape(B, n_sp * T, D)
    res_t = linear(res_t, p, name + ".temporal_fc")
    xt = x[:, 1:, :] + res_t
    # spatial attention per frame with the CLS token replicated per frame (vit.py:165-181)
    init_cls = x[:, 0, :].unsqueeze(1)
    cls_rep = init_cls.repeat(1, T, 1).reshape(B * T, 1, D)
    xs = xt.reshape(B, n_sp, T, D).permute(0, 2, 1, 3).reshape(B * T, n_sp, D)
    xs = torch.cat((cls_rep, xs), 1)
    res_s = vit_attention(layer_norm(xs, p, name + ".norm1", VIT_EPS), p, name + ".attn")
    res_s = _row_scale(res_s, drop.get("s"))
    # CLS averaged over frames, patches scattered back to (h w t) order (vit.py:184-196)
    cls_tok = res_s[:, 0, :].reshape(B, T, D).mean(1, keepdim=True)
    res = res_s[:, 1:, :].reshape(B, T, n_sp, D).permute(0, 2, 1, 3).reshape(B, n_sp * T, D)
    x = torch.cat((init_cls, xt), 1) + torch.cat((cls_tok, res), 1)
    # MLP (vit.py:198-212, Mlp.forward vit.py:59-65)
    h = layer_norm(x, p, name + ".norm2", VIT_EPS)
    h = linear(gelu_erf(linear(h, p, name + ".mlp.fc1")), p, name + ".mlp.fc2")
    return x + _row_scale(h, drop.get("m"))


def patch_embed(x_bcthw, p, name, patch=16):
    """vit.py:233-239: (b c t h w) -> ((b t), h/16*w/16, D) via a stride-16 conv."""
    B, C, T, H, W = x_bcthw.shape
    x = x_bcthw.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    x = F.conv2d(x, p[name + ".proj.weight"], p[name + ".proj.bias"], stride=patch)
    Wp = x.size(-1)
    return x.flatten(2).transpose(1, 2), T, Wp


def vit_forward_features(x_bcthw, p, name, drops=None, depth=12):
    """vit.py:321-377 (VisionTransformer.forward_features, return_all_tokens=True).

    Assumes pos_embed / time_embed match the input grid (the resize branches vit.py:328-340,
    351-356 are checkpoint-loading conveniences outside the hot path)."""
    B = x_bcthw.shape[0]
    x, T, W = patch_embed(x_bcthw, p, name + ".patch_embed")
    D = x.size(-1)
    cls = p[name + ".cls_token"].expand(x.size(0), -1, -1)
    x = torch.cat((cls, x), dim=1) + p[name + ".pos_embed"]
    cls_tokens = x[:B, 0, :].unsqueeze(1)
    n = x.size(1) - 1
    x = x[:, 1:].reshape(B, T, n, D).permute(0, 2, 1, 3).reshape(B * n, T, D)
    x = x + p[name + ".time_embed"]
    x = x.reshape(B, n * T, D)
    x = torch.cat((cls_tokens, x), dim=1)
    for i in range(depth):
        x = vit_block(x, p, "%s.blocks.%d" % (name, i), B, T, W, None if drops is None else drops[i])
    return layer_norm(x, p, name + ".norm", VIT_EPS)


def timesformer_forward_features(x_bcthw, p, name, num_frm, drops=None):
    """vit.py:475-503 (TimeSformer.forward_features, pooling='temporal') -> (B, 1+N, D)."""
    x = vit_forward_features(x_bcthw, p, name + ".model", drops)
    B, _, D = x.shape
    cls = x[:, 0, :].unsqueeze(1)
    other = x[:, 1:, :].reshape(B, -1, num_frm, D).mean(dim=2)  # 'b (h w t) m -> b t (h w) m', mean over t
    return torch.cat((cls, other), dim=1)


# ----------------------------------------------------------------------------- BERT (xbert.py)
def bert_embeddings(input_ids, p, name, eps):
    """xbert.py:186-213 (token_type 0, absolute positions, eval dropout)."""
    L = input_ids.shape[1]
    e = p[name + ".word_embeddings.weight"][input_ids]
    e = e + p[name + ".token_type_embeddings.weight"][0]
    e = e + p[name + ".position_embeddings.weight"][:L]
    return layer_norm(e, p, name + ".LayerNorm", eps)


def extended_attention_mask(mask):
    """xbert.py:878-938 (encoder branch): (B, L) {0,1} -> additive (B,1,1,L)."""
    return (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0


def bert_layer(h, ext_mask, p, name, heads, eps):
    """xbert.py:457-519: self-attention (263-346) + SelfOutput (356-360) + FFN (421-438)."""
    B, L, C = h.shape
    hd = C // heads

    def split(t):
        return t.view(B, L, heads, hd).permute(0, 2, 1, 3)

    q = split(linear(h, p, name + ".attention.self.query"))
    k = split(linear(h, p, name + ".attention.self.key"))
    v = split(linear(h, p, name + ".attention.self.value"))
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    s = s + ext_mask
    pr = s.softmax(dim=-1)
    ctx = torch.matmul(pr, v).permute(0, 2, 1, 3).reshape(B, L, C)
    a = layer_norm(linear(ctx, p, name + ".attention.output.dense") + h, p, name + ".attention.output.LayerNorm", eps)
    i = gelu_erf(linear(a, p, name + ".intermediate.dense"))
    return layer_norm(linear(i, p, name + ".output.dense") + a, p, name + ".output.LayerNorm", eps)


def bert_model(p, name, cfg, attention_mask, input_ids=None, encoder_embeds=None, mode="text"):
    """xbert.py:940-1081 (BertModel.forward) + :549-559 layer ranges.

    mode='text'   : embeddings + layers [0, fusion_layer)
    mode='fusion' : encoder_embeds bypass the embeddings (xbert.py:1044-1053), layers [fusion_layer, L)
    """
    ext = extended_attention_mask(attention_mask)
    if encoder_embeds is None:
        h = bert_embeddings(input_ids, p, name + ".embeddings", cfg["layer_norm_eps"])
    else:
        h = encoder_embeds
    lo, hi = (0, cfg["fusion_layer"]) if mode == "text" else (cfg["fusion_layer"], cfg["num_hidden_layers"])
    for i in range(lo, hi):
        h = bert_layer(h, ext, p, "%s.encoder.layer.%d" % (name, i), cfg["num_attention_heads"], cfg["layer_norm_eps"])
    return h


def mlm_head(h, p, name, eps):
    """xbert.py:665-682: dense + GELU + LN, tied decoder + bias."""
    t = layer_norm(gelu_erf(linear(h, p, name + ".transform.dense")), p, name + ".transform.LayerNorm", eps)
    return F.linear(t, p[name + ".decoder.weight"], p[name + ".bias"])


# ----------------------------------------------------------------------------- ALPRO heads
def default_neg_sampler(weights):
    """Deterministic stand-in for torch.multinomial(weights[b], 1) (alpro_models.py:303,311)."""
    return weights.argmax(dim=1)


class AlproOracle:
    """Functional mirror of AlproBaseModel/AlproForPretrain/Prompter/AlproForVideoTextRetrieval.

    p: dict with the reference's state_dict keys (tied aliases may be absent; they are filled in).
    bert_cfg: dict of config_release/base_model.json; num_frm from the task config.
    world: (rank, [video_feat per rank], [text_feat per rank]) to emulate hvd.allgather, else None.
    """

    def __init__(self, p, bert_cfg, num_frm, prefix=""):
        self.p = dict(p)
        self.cfg = dict(bert_cfg)
        self.T = num_frm
        self.pre = prefix
        te = prefix + "text_encoder."
        if te + "bert.embeddings.word_embeddings.weight" in self.p:
            self.p.setdefault(te + "cls.predictions.decoder.weight", self.p[te + "bert.embeddings.word_embeddings.weight"])
        if te + "cls.predictions.bias" in self.p:
            self.p.setdefault(te + "cls.predictions.decoder.bias", self.p[te + "cls.predictions.bias"])

    # -- encoders ---------------------------------------------------------------------
    def temp(self):
        return self.p[self.pre + "temp"].clamp(0.001, 0.5)  # alpro_models.py:80-81

    def visual_embeds(self, visual_inputs, drops=None):
        """alpro_models.py:186-194: (B,T,C,H,W) -> transpose -> TimeSformer.forward_features."""
        return timesformer_forward_features(visual_inputs.transpose(1, 2), self.p, self.pre + "visual_encoder", self.T, drops)

    def text_embeds(self, ids, mask):
        return bert_model(self.p, self.pre + "text_encoder.bert", self.cfg, mask, input_ids=ids, mode="text")

    def fusion(self, embeds, mask):
        return bert_model(self.p, self.pre + "text_encoder.bert", self.cfg, mask, encoder_embeds=embeds, mode="fusion")

    def video_feat(self, video_embeds):
        return F.normalize(linear(video_embeds[:, 0, :], self.p, self.pre + "vision_proj"), dim=-1)

    def text_feat(self, text_embeds):
        return F.normalize(linear(text_embeds[:, 0, :], self.p, self.pre + "text_proj"), dim=-1)

    # -- VTC (alpro_models.py:109-128 / 564-587 / 763-779) -------------------------------
    def vtc(self, video_feat, text_feat, world=None):
        b = video_feat.shape[0]
        if world is None:
            rank, gv, gt = 0, video_feat, text_feat
        else:
            rank, vs, ts = world
            vs, ts = list(vs), list(ts)
            vs[rank], ts[rank] = video_feat, text_feat
            gv, gt = torch.cat(vs, 0), torch.cat(ts, 0)
        t = self.temp()
        sim_v2t = video_feat @ gt.t() / t
        sim_t2v = text_feat @ gv.t() / t
        targets = torch.zeros_like(sim_v2t)
        targets[:, b * rank: b * (rank + 1)] = torch.eye(b)
        loss_v2t = -torch.sum(F.log_softmax(sim_v2t, dim=1) * targets, dim=1).mean()
        loss_t2v = -torch.sum(F.log_softmax(sim_t2v, dim=1) * targets, dim=1).mean()
        return (loss_v2t + loss_t2v) / 2, sim_v2t, sim_t2v, targets, rank

    # -- VTM (alpro_models.py:269-344 / 800-872) ------------------------------------------
    def vtm(self, text_embeds, text_atts, video_embeds, sim_v2t, sim_t2v, rank=0, neg_sampler=default_neg_sampler):
        bs = text_embeds.shape[0]
        video_atts = torch.ones(video_embeds.shape[:-1], dtype=torch.long)
        pos = self.fusion(torch.cat([text_embeds, video_embeds], 1), torch.cat([text_atts, video_atts], 1))
        with torch.no_grad():
            w_i2t = sim_v2t[:, bs * rank: bs * (rank + 1)].clone()
            w_t2i = sim_t2v[:, bs * rank: bs * (rank + 1)].clone()
            w_i2t.fill_diagonal_(NEG_INF)
            w_t2i.fill_diagonal_(NEG_INF)
            w_i2t, w_t2i = F.softmax(w_i2t, dim=1), F.softmax(w_t2i, dim=1)
            neg_v = neg_sampler(w_t2i)  # a negative video for each text
            neg_t = neg_sampler(w_i2t)  # a negative text for each video
        t_all = torch.cat([text_embeds, text_embeds[neg_t]], 0)
        ta_all = torch.cat([text_atts, text_atts[neg_t]], 0)
        v_all = torch.cat([video_embeds[neg_v], video_embeds], 0)
        va_all = torch.cat([video_atts, video_atts], 0)
        neg = self.fusion(torch.cat([t_all, v_all], 1), torch.cat([ta_all, va_all], 1))
        vl = torch.cat([pos[:, 0, :], neg[:, 0, :]], 0)
        logits = linear(vl, self.p, self.pre + "itm_head")
        labels = torch.cat([torch.ones(bs, dtype=torch.long), torch.zeros(2 * bs, dtype=torch.long)])
        return F.cross_entropy(logits, labels), logits, labels, pos

    # -- MLM (alpro_models.py:346-373) -----------------------------------------------------
    def mlm(self, mlm_ids, text_mask, video_embeds, mlm_labels):
        te = self.text_embeds(mlm_ids, text_mask)
        video_atts = torch.ones(video_embeds.shape[:-1], dtype=torch.long)
        out = self.fusion(torch.cat([te, video_embeds], 1), torch.cat([text_mask, video_atts], 1))
        Lt = text_mask.shape[1]
        logits = mlm_head(out[:, :Lt], self.p, self.pre + "text_encoder.cls.predictions", self.cfg["layer_norm_eps"])
        loss = F.cross_entropy(logits.view(-1, self.cfg["vocab_size"]), mlm_labels.view(-1))
        return loss, logits

    # -- MPM / PEM (alpro_models.py:209-232, 525-551) ---------------------------------------
    def pseudo_labels(self, crop_visual_inputs, kind="video"):
        """Prompter.get_pseudo_labels (alpro_models.py:531-551); self must be a prompter-prefixed oracle."""
        with torch.no_grad():
            feat = self.video_feat(self.visual_embeds(crop_visual_inputs))
            prompt = self.p[self.pre + ("video_prompt_feat" if kind == "video" else "image_prompt_feat")]
            sim = feat @ prompt.t() / self.p[self.pre + "temp"]  # NB: prompter temp is NOT clamped here (:547)
            soft = sim.softmax(dim=1)
            ignore = torch.max(sim, dim=1)[1] < 0.2  # quirk: compares the argmax INDEX (alpro_models.py:527)
        return soft, ignore

    def mpm(self, fusion_pos, Lt, soft_labels, ignore, patch_masks):
        vis = fusion_pos[:, Lt + 1:]
        bsz = patch_masks.shape[0]
        inv = (1 - patch_masks.view(bsz, -1)).unsqueeze(-1)
        n_masked = inv.squeeze(-1).sum(-1, keepdim=True)
        emb = (inv * vis).sum(1) / n_masked
        h = F.relu(linear(emb, self.p, self.pre + "mpm_head.0"))
        logits = linear(h, self.p, self.pre + "mpm_head.2")
        ce = -torch.sum(F.log_softmax(logits, dim=1) * soft_labels, dim=1)
        ce = torch.where(ignore, torch.zeros_like(ce), ce)
        return ce.sum() / (bsz - ignore.sum()), logits

    # -- top-level forwards -----------------------------------------------------------------
    def forward_pretrain(self, batch, world=None, neg_sampler=default_neg_sampler):
        """AlproForPretrain.forward (alpro_models.py:79-183), use_mask_prob == 0 branch."""
        ve = self.visual_embeds(batch["visual_inputs"])
        vf = self.video_feat(ve)
        te = self.text_embeds(batch["text_input_ids"], batch["text_input_mask"])
        tf = self.text_feat(te)
        itc, s_v2t, s_t2v, _, rank = self.vtc(vf, tf, world)
        itm, itm_logits, itm_labels, pos = self.vtm(te, batch["text_input_mask"], ve, s_v2t, s_t2v, rank, neg_sampler)
        out = dict(itc_loss=itc, itm_loss=itm, itm_scores=itm_logits, itm_labels=itm_labels,
                   video_embeds=ve, video_feat=vf, text_embeds=te, text_feat=tf, sim_v2t=s_v2t, sim_t2v=s_t2v,
                   mlm_scores=None, mlm_loss=None, mlm_labels=None, mpm_loss=None, mpm_logits=None, mpm_labels=None)
        if "mlm_labels" in batch:
            out["mlm_loss"], out["mlm_scores"] = self.mlm(batch["mlm_text_input_ids"], batch["text_input_mask"], ve, batch["mlm_labels"])
            out["mlm_labels"] = batch["mlm_labels"]
        if "mpm_mask" in batch:
            prompter = AlproOracle(self.p, self.cfg, self.T, prefix=self.pre + "prompter.")
            soft, ignore = prompter.pseudo_labels(batch["crop_visual_inputs"], batch["type"])
            out["mpm_loss"], out["mpm_logits"] = self.mpm(pos, batch["text_input_mask"].shape[1], soft, ignore, batch["mpm_mask"])
            out["mpm_labels"] = soft
        return out

    def forward_retrieval(self, batch, world=None, neg_sampler=default_neg_sampler):
        """AlproForVideoTextRetrieval.forward (alpro_models.py:733-798)."""
        ve = self.visual_embeds(batch["visual_inputs"])
        vf = self.video_feat(ve)
        te = self.text_embeds(batch["text_input_ids"], batch["text_input_mask"])
        tf = self.text_feat(te)
        itc, s_v2t, s_t2v, _, rank = self.vtc(vf, tf, world)
        itm, logits, labels, _ = self.vtm(te, batch["text_input_mask"], ve, s_v2t, s_t2v, rank, neg_sampler)
        return dict(itc_loss=itc, itm_loss=itm, itm_scores=logits, itm_labels=labels)

    def forward_inference(self, batch):
        """AlproForVideoTextRetrieval.forward_inference (alpro_models.py:874-914): 1 video x n captions."""
        ve = self.visual_embeds(batch["visual_inputs"])
        vf = self.video_feat(ve)
        n = batch["text_input_mask"].shape[0]
        ve = ve.repeat(n, 1, 1)
        te = self.text_embeds(batch["text_input_ids"], batch["text_input_mask"])
        tf = self.text_feat(te)
        itc_scores = vf @ tf.t() / self.p[self.pre + "temp"]
        video_atts = torch.ones(ve.shape[:-1], dtype=torch.long)
        out = self.fusion(torch.cat([te, ve], 1), torch.cat([batch["text_input_mask"], video_atts], 1))
        return dict(logits=linear(out[:, 0, :], self.p, self.pre + "itm_head"), itc_scores=itc_scores)

    def build_text_prompts(self, ids, mask, entity_num, step_size=10000):
        """One prompt set of Prompter.build_text_prompts (alpro_models.py:430-507): text-encode the E x n_templates prompt
        sentences in chunks of step_size, project + normalise the CLS rows, average over the templates (the encoded list is
        template-major: chunk(n_templates) yields one (E, 256) block per template, :470-472)."""
        feats = []
        with torch.no_grad():
            for s in range(0, ids.shape[0], step_size):
                feats.append(self.text_feat(self.text_embeds(ids[s:s + step_size], mask[s:s + step_size])))
            feat = torch.cat(feats, 0)
            n_templates = int(feat.shape[0] / entity_num)
            return torch.stack(feat.chunk(n_templates), dim=1).mean(dim=1)

    def forward_prompter(self, batch, world=None):
        """Prompter.forward (alpro_models.py:553-594)."""
        vf = self.video_feat(self.visual_embeds(batch["visual_inputs"]))
        tf = self.text_feat(self.text_embeds(batch["text_input_ids"], batch["text_input_mask"]))
        itc, s_v2t, s_t2v, targets, _ = self.vtc(vf, tf, world)
        return dict(itc_loss=itc, itc_labels=targets.max(dim=1)[1],
                    i2t_scores=F.log_softmax(s_v2t, dim=1), t2i_scores=F.log_softmax(s_t2v, dim=1))


# ----------------------------------------------------------------------------- state-dict spec
def alpro_state_spec(kind, bert_cfg, num_frm, img_size=224, num_entities=1000, prefix=""):
    """Ordered {state_dict key: shape} for kind in {'retrieval','prompter','pretrain'}."""
    D, H = 768, bert_cfg["hidden_size"]
    n = (img_size // 16) ** 2
    spec = {}

    def lin(name, out, inp):
        spec[prefix + name + ".weight"] = (out, inp)
        spec[prefix + name + ".bias"] = (out,)

    def ln(name, d):
        spec[prefix + name + ".weight"] = (d,)
        spec[prefix + name + ".bias"] = (d,)

    spec[prefix + "temp"] = ()
    v = "visual_encoder.model."
    spec[prefix + v + "cls_token"] = (1, 1, D)
    spec[prefix + v + "pos_embed"] = (1, n + 1, D)
    spec[prefix + v + "time_embed"] = (1, num_frm, D)
    spec[prefix + v + "patch_embed.proj.weight"] = (D, 3, 16, 16)
    spec[prefix + v + "patch_embed.proj.bias"] = (D,)
    for i in range(12):
        b = v + "blocks.%d." % i
        ln(b + "norm1", D); lin(b + "attn.qkv", 3 * D, D); lin(b + "attn.proj", D, D)
        ln(b + "temporal_norm1", D); lin(b + "temporal_attn.qkv", 3 * D, D); lin(b + "temporal_attn.proj", D, D)
        lin(b + "temporal_fc", D, D); ln(b + "norm2", D); lin(b + "mlp.fc1", 4 * D, D); lin(b + "mlp.fc2", D, 4 * D)
    ln(v + "norm", D); lin(v + "head", 400, D)
    t = "text_encoder.bert."
    spec[prefix + t + "embeddings.position_ids"] = (1, bert_cfg["max_position_embeddings"])
    spec[prefix + t + "embeddings.word_embeddings.weight"] = (bert_cfg["vocab_size"], H)
    spec[prefix + t + "embeddings.position_embeddings.weight"] = (bert_cfg["max_position_embeddings"], H)
    spec[prefix + t + "embeddings.token_type_embeddings.weight"] = (bert_cfg["type_vocab_size"], H)
    ln(t + "embeddings.LayerNorm", H)
    for i in range(bert_cfg["num_hidden_layers"]):
        l = t + "encoder.layer.%d." % i
        lin(l + "attention.self.query", H, H); lin(l + "attention.self.key", H, H); lin(l + "attention.self.value", H, H)
        lin(l + "attention.output.dense", H, H); ln(l + "attention.output.LayerNorm", H)
        lin(l + "intermediate.dense", bert_cfg["intermediate_size"], H)
        lin(l + "output.dense", H, bert_cfg["intermediate_size"]); ln(l + "output.LayerNorm", H)
    c = "text_encoder.cls.predictions."
    spec[prefix + c + "bias"] = (bert_cfg["vocab_size"],)
    lin(c + "transform.dense", H, H); ln(c + "transform.LayerNorm", H)
    spec[prefix + c + "decoder.weight"] = (bert_cfg["vocab_size"], H)
    spec[prefix + c + "decoder.bias"] = (bert_cfg["vocab_size"],)
    lin("vision_proj", 256, D); lin("text_proj", 256, H); lin("itm_head", 2, H)
    if kind == "prompter":
        spec[prefix + "video_prompt_feat"] = (num_entities, 256)
        spec[prefix + "image_prompt_feat"] = (num_entities, 256)
    if kind == "pretrain":
        spec.update(alpro_state_spec("prompter", bert_cfg, num_frm, img_size, num_entities, prefix + "prompter."))
        lin("mpm_head.0", 2 * H, H); lin("mpm_head.2", num_entities, 2 * H)
    return spec


def det_state(kind, bert_cfg, num_frm, img_size=224, num_entities=1000, only=None):
    """Closed-form weights for the whole model (oracle/det_init.py), keyed like the reference."""
    from oracle.det_init import canonical_name, det_param
    p = {}
    for k, shape in alpro_state_spec(kind, bert_cfg, num_frm, img_size, num_entities).items():
        if only is not None and not any(k.startswith(o) for o in only):
            continue
        ck = canonical_name(k)
        p[k] = p[ck] if ck in p and ck != k else det_param(ck, shape)
    return p


# ----------------------------------------------------------------------------- optimizer epilogue
def warmup_linear(step, warmup_step, tot_step):
    """src/optimization/sched.py:14-17."""
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def lr_sched(global_step, decay, learning_rate, num_train_steps, warmup_ratio=0.1):
    """src/optimization/sched.py:28-49 (get_lr_sched), the 'linear' and 'constant' branches the release configs use."""
    warmup_steps = int(warmup_ratio * num_train_steps)
    if decay == "linear":
        lr = learning_rate * warmup_linear(global_step, warmup_steps, num_train_steps)
    elif decay == "constant":
        lr = learning_rate
    else:
        raise ValueError(decay)
    return lr if lr > 0 else 1e-8


def clip_and_adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, max_norm=None,
                        correct_bias=True):
    """One optimizer step of the reference's loop on lists of tensors, in place: torch.nn.utils.clip_grad_norm_
    (run_pretrain_sparse.py:633: total 2-norm over all gradients, coefficient max_norm / (total + 1e-6) clamped to 1) followed by
    AdamW.step (src/optimization/adamw.py:77-101: moments, bias-corrected step size, update, decoupled weight decay with the step's lr).
    `step` counts from 1.  Returns the total gradient norm before clipping (what the driver logs)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = 1.0
    if max_norm is not None and max_norm > 0:
        coef = min(float(max_norm) / (float(total) + 1e-6), 1.0)
    b1, b2 = betas
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        g = g * coef
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)
        if weight_decay > 0.0:
            p.add_(p, alpha=-lr * weight_decay)
    return float(total)
