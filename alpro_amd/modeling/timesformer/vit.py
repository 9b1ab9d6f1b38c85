"""TimeSformer (divided space-time ViT-B/16) on the MI355X kernels.

Mirrors the reference interface of src/modeling/timesformer/vit.py -- class names, constructor
arguments, `forward_features` signatures and the state_dict layout (`model.blocks.{i}.attn.qkv.weight`
...) -- while the arithmetic runs in libalpro_hip.so:

  reference (vit.py)                                   here
  ---------------------------------------------------  -----------------------------------------------
  PatchEmbed conv + flatten (:233-239), +pos/+time     alpro_patchify + alpro_gemm(PATCH_EMBED map) whose
  embeds via two rearranges (:342-361)                 residual is a (N*T, D) table bias+pos+time
  rearrange 'b (h w t) m -> (b h w) t m' (:147)        LayerNorm with SKIP_CLS gather (no copy)
  Attention (:81-100) on (B*N, T, D)                   alpro_gemm(qkv) -> alpro_attn_temporal -> alpro_gemm(proj)
  temporal_fc + residual (:161-162), rearrange/cat     the projection GEMM writes its 16-bit output only; the residual add,
  to (B*T, 1+N, D) (:165-172), norm1 (:180)            the frame-token gather and norm1 are ONE streaming kernel
                                                       (alpro_add_layernorm_fwd, PRE_SPATIAL)
  Attention on (B*T, 1+N, D) (:180)                    alpro_gemm(qkv) -> alpro_attn -> alpro_gemm(proj), 16-bit output
  CLS mean + scatter back + residual (:184-196),       alpro_add_layernorm_fwd (PRE_MLP): scatter map, frame mean of the CLS
  norm2 (:200)                                         rows, residual add and norm2 in one pass
  norm2 + Mlp + residual (:198-212)                    LayerNorm -> alpro_gemm(GELU) -> alpro_gemm(residual)
  norm (:372) + temporal mean pool (:484-492)          alpro_vit_final_pool
  DropPath per (b n)/(b t)/b rows (vit_utils.py:137)   row_scale vector in the GEMM epilogue
"""
import math
import os
from functools import partial

import torch
import torch.nn as nn

from alpro_amd import config as rt
from alpro_amd import hip
from alpro_amd.modeling import train as tr
from alpro_amd.modeling.weights import OperandCache, param_version

VIT_EPS = 1e-6


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """vit_utils.py:56-76 semantics (delegates to torch's identical implementation)."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class DropPath(nn.Module):
    """Stochastic depth (vit_utils.py:154-162).  In the fused path the Bernoulli row mask is sampled by
    `row_scale` below and applied inside the producing GEMM's epilogue."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def row_scale(self, rows, device):
        if not self.drop_prob or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        return torch.floor(keep + torch.rand(rows, dtype=torch.float32, device=device)) / keep


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., with_qkv=True):
        super().__init__()
        assert with_qkv, "the ALPRO path always uses with_qkv=True (vit.py:114,120)"
        assert attn_drop == 0. and proj_drop == 0., "attn_drop/proj_drop are 0 in every release config"
        self.num_heads = num_heads
        head_dim = dim // num_heads
        assert head_dim == 64, "kernels are specialised for head_dim 64 (ViT-B/16, BERT-base)"
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _ClsSide:
    """The precise-CLS chain of the ViT blocks on a second stream (alpro_amd.config.cls_stream, round 5).

    Per block the chain is: LayerNorm + q | k | v of the B CLS rows (needs only the block INPUT's CLS rows: the temporal half never touches
    them) -> [the CLS query's attention, inside the main path's attention launch] -> projection of B*T rows -> frame mean + residual ->
    norm2 + fc1 + GELU -> fc2 + residual, written over the 16-bit path's CLS rows of the block output.  Five skinny launches of 10-35 us
    that each leave most of the chip idle when they sit BETWEEN the block's big launches.  Here they go to a side stream:
        side:  [wait: block input ready]  snapshot x[:, 0], q|k|v rows                                   -> ev_q
        main:  temporal half, add+norm1, qkv GEMM, [wait ev_q] attention (+ CLS query)                   -> ev_attn, ... fc2 -> ev_out
        side:  [wait ev_attn / the main path's x2] proj rows, frame mean, fc1 rows, [wait ev_out] fc2 rows over out[:, 0]  -> done
        main (next block):  [wait done] before its first read of the CLS rows (the add + norm1 kernel)
    so the first part runs beside the temporal half and the second beside the MLP GEMMs.  All side-stream outputs live in buffers owned by
    this object (no allocation crosses streams); the event order above is also what makes their re-use from block to block safe."""
    _by_device = {}

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device)
        self.bufs = {}
        self.done = None

    @classmethod
    def _key(cls, device):
        # one chain (side stream + buffers) per LAUNCH stream: the two-stream block runner (run_blocks) drives two of them at once
        return (device.type, device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)

    @classmethod
    def get(cls, device):
        key = cls._key(device)
        obj = cls._by_device.get(key)
        if obj is None:
            obj = cls._by_device[key] = cls(device)
        return obj

    def buf(self, name, shape, device):
        t = self.bufs.get((name, shape))
        if t is None:
            t = self.bufs[(name, shape)] = torch.empty(shape, dtype=torch.float32, device=device)
        return t

    def wait_done(self):
        """Main stream: the previous block's chain has written its CLS rows."""
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)
            self.done = None

    @classmethod
    def join(cls, device):
        """The current stream waits for the chain that was driven from it."""
        obj = cls._by_device.get(cls._key(device)) if device.type == "cuda" else None
        if obj is not None:
            obj.wait_done()


_AUX_STREAMS = {}


def _aux_stream(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    s = _AUX_STREAMS.get(key)
    if s is None:
        s = _AUX_STREAMS[key] = torch.cuda.Stream(device)
    return s


def run_blocks(blocks, tok, B, T, W):
    """The no-grad forward through `blocks` (tok updated in place and returned), CLS chains joined.

    Round 6: with alpro_amd.config.split_streams(B) the two halves of the batch go through every block on two HIP streams (clips are
    independent: vit.py:146-212 never mixes them).  Each half's launches are ordered on its own stream (and its precise-CLS chain on that
    stream's own side stream); the second stream is forked behind the embedding -- and again behind any block whose operand copies the launch
    stream had to (re)build -- and the launch stream waits for it once, behind the last block (ALPRO_SPLIT_LOCKSTEP=1: at every block boundary,
    a measurement aid that costs more than the split returns).  What it buys: a persistent GEMM's last, partly filled round of tiles and every
    kernel's ramp-up / drain no longer idle the chip -- workgroups of the other half's next launch take the CUs as they are vacated
    (B = 32 x 8 frames: -1.3 ... -3.2 % per forward, profiles/r6_split_streams_ab.txt).  Outputs are those of the one-stream forward bit for
    bit as long as the half batch sends every Linear to the same GEMM kernel as the whole batch."""
    dev = tok.device
    if not (tok.is_cuda and not torch.is_grad_enabled() and rt.split_streams(B)):
        for blk in blocks:
            tok = blk(tok, B, T, W)
        _ClsSide.join(dev)
        return tok
    from alpro_amd.modeling import weights
    main, aux = torch.cuda.current_stream(dev), _aux_stream(dev)
    h = B // 2
    lo, hi = tok[:h], tok[h:]
    lockstep = rt.split_lockstep()
    ev = main.record_event()        # the embedding is written
    for i, blk in enumerate(blocks):
        r0 = weights.operand_rebuilds()
        blk(lo, h, T, W)
        if weights.operand_rebuilds() != r0:   # the launch stream (re)built operand copies of this block: the second stream may only read them afterwards
            ev = main.record_event()
        with torch.cuda.stream(aux):
            if ev is not None:
                aux.wait_event(ev)
                ev = None
            blk(hi, h, T, W)
            if lockstep or i == len(blocks) - 1:
                _ClsSide.join(dev)
                ev_hi = aux.record_event()
        if lockstep or i == len(blocks) - 1:
            _ClsSide.join(dev)
            main.wait_event(ev_hi)
            if lockstep:
                ev = main.record_event()
    return tok


class Block(nn.Module):
    def __init__(self, dim, num_heads, layer_num, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0.1, act_layer=nn.GELU, norm_layer=nn.LayerNorm, attention_type='divided_space_time',
                 use_grad_checkpointing=False):
        super().__init__()
        assert attention_type == 'divided_space_time', "TimeSformer.__init__ hard-codes divided_space_time (vit.py:435)"
        self.attention_type = attention_type
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.temporal_norm1 = norm_layer(dim)
        self.temporal_attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.temporal_fc = nn.Linear(dim, dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.layer_num = layer_num
        self.use_grad_checkpointing = use_grad_checkpointing
        self._ops = OperandCache()

    def _w(self, name, lin, dt):
        return self._ops.get(name, lin.weight, dt)

    # ---- merged temporal projection ------------------------------------------------------------------------------
    # vit.py:157-162 applies two Linears back to back with only drop_path between them:
    #     res_temporal = drop_path(temporal_attn.proj(attn_out));  xt = x + temporal_fc(res_temporal)
    # drop_path is a per-row scale s, so   temporal_fc(s * (a Wp^T + bp)) = s * (a (Wfc Wp)^T + Wfc bp) + bfc :
    # ONE (768 x 768) GEMM with the merged weight We = Wfc Wp (rebuilt per optimizer step from the two fp32 parameters, a
    # 0.9 GFLOP product), bias Wfc bp under the row scale and bfc after it (alpro_gemm bias2).  Saves one 100k x 768 x 768
    # GEMM per block in forward and, in backward, one dgrad and one wgrad: dWe = (s*dY)^T a is taken once and pushed through
    # the product rule (dWfc = dWe Wp^T + db1 bp^T, dWp = Wfc^T dWe, dbp = Wfc^T db1, dbfc = colsum(dY)).
    merge_temporal_proj = True
    fuse_ln_bwd_emit = os.environ.get("ALPRO_FUSE_LN_BWD", "1") != "0"   # LayerNorm backward also emits the next GEMMs' operand rows (alpro_layernorm_bwd_emit) instead of a gather_cast pass; 0 = round-2 form (A/B)
    fuse_residual_ln = os.environ.get("ALPRO_FUSE_RESIDUAL_LN", "1") != "0"   # residual adds of the two attention halves inside the following LayerNorm (alpro_add_layernorm_fwd); 0 = round-2 form (A/B measurements)

    batch_merged_tproj = os.environ.get("ALPRO_BATCH_TPROJ", "1") != "0"   # the 12 blocks' merged-projection upkeep in batched launches (_MergedTProjBank); 0 = per block (A/B)

    def _bank(self):
        bank = getattr(self, "_bank_obj", None) if (self.batch_merged_tproj and self.merge_temporal_proj) else None
        if bank is not None and (len(bank.blocks) <= self._bank_idx or bank.blocks[self._bank_idx] is not self):   # a deep copy of the model still points at the original's bank
            return None
        return bank

    def _merged_tproj(self, dt):
        bank = self._bank()
        if bank is not None:
            return bank.get(self._bank_idx, dt)
        wp, bp, wf = self.temporal_attn.proj.weight, self.temporal_attn.proj.bias, self.temporal_fc.weight
        ver = (param_version(wp), param_version(bp), param_version(wf), dt)
        hit = self._ops._store.get("t_merged")
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            we = hip.gemm(wf.detach().contiguous(), hip.transpose(wp.detach().contiguous()), out_dtype=torch.float32)   # (768, 768) fp32 = Wfc Wp
            b1 = torch.mv(wf.detach(), bp.detach()).contiguous()
            m = dict(w=we if dt == torch.float32 else hip.cast(we, dt), wT=hip.transpose(we, out_dtype=dt, pad_to=64), b1=b1)
        self._ops._store["t_merged"] = (ver, m)
        from alpro_amd.modeling import weights
        weights.note_operand_rebuild()
        return m

    # ---- precise CLS rows (alpro_amd.config.cls_precise; csrc/cls_precise.hip) -------------------------------------------------
    # The CLS row of the block output, re-evaluated in fp32 from the block input's CLS row: LayerNorm -> q | k | v (vit.py:84-85 on the
    # row that :165-167 replicates per frame: ONE row per clip) -> the CLS query's attention over the frame's tokens (K / V of the patch
    # tokens as the 16-bit GEMM produced them) -> proj -> frame mean + residual (:184-187) -> norm2 -> Mlp -> residual (:198-212).
    # B*T + 3*B fp32 rows per block against B*(1 + N*T) 16-bit rows; drop-path scales are the main path's.
    def _cls_qkv(self, x_cls_in, out=None):
        """(B, D) fp32 CLS rows of the block input -> their unrounded q in a (B, 3 D) buffer laid out like q | k | v (norm1 fused into the operand
        load).  Only the q third is computed (round 6): alpro_attn_fwd's CLS query reads q from this buffer and every K / V row -- the CLS
        token's included -- from the 16-bit images in LDS, so the k and v thirds were 2/3 of a 25 us launch per block for nothing."""
        sa = self.attn
        D = x_cls_in.shape[-1]
        if out is None:
            out = torch.empty((x_cls_in.shape[0], 3 * D), dtype=torch.float32, device=x_cls_in.device)
        hip.gemm_rows(x_cls_in, self._w("s_qkv", sa.qkv, torch.float32)[:D], bias=sa.qkv.bias[:D], ln=(self.norm1.weight, self.norm1.bias, VIT_EPS), out=out[:, :D])
        return out

    def _cls_side_begin(self, side, x, B, T, snapshot):
        """Side stream, at block entry: (snapshot of) the block input's CLS rows and their q | k | v.  -> (x_cls_in, cls_q, o_c buffer)."""
        main = torch.cuda.current_stream()
        D = x.shape[-1]
        ev_in = main.record_event()
        with torch.cuda.stream(side.stream):
            side.stream.wait_event(ev_in)
            if snapshot:    # the inference path updates x in place
                x_cls_in = side.buf("x_cls_in", (B, D), x.device)
                x_cls_in.copy_(x[:, 0])
            else:
                x_cls_in = x[:, 0]
            cls_q = self._cls_qkv(x_cls_in, out=side.buf("cls_q", (B, 3 * D), x.device))
            side.ev_q = side.stream.record_event()
        return x_cls_in, cls_q, side.buf("o_c", (B * T, D), x.device)

    def _cls_side_finish(self, side, ev_mid, ev_out, x_cls_in, o_c, B, T, drop_s, drop_m, x2_out, out_out):
        """Side stream: the chain behind the attention (ev_mid: o_c -- and, training, the main path's x2 -- are written; ev_out: so is the block output)."""
        D = o_c.shape[-1]
        sa, f32, dev = self.attn, torch.float32, o_c.device
        with torch.cuda.stream(side.stream):
            side.stream.wait_event(ev_mid)
            p_c = hip.gemm_rows(o_c, self._w("s_proj", sa.proj, f32), bias=sa.proj.bias, row_scale=drop_s, out=side.buf("p_c", (B * T, D), dev))
            hip.cls_mean_residual(x_cls_in, p_c, x2_out, B, T)
            f1c = hip.gemm_rows(x2_out, self._w("fc1", self.mlp.fc1, f32), bias=self.mlp.fc1.bias, act=hip.ACT_GELU, ln=(self.norm2.weight, self.norm2.bias, VIT_EPS),
                                out=side.buf("f1c", (B, self.mlp.fc1.out_features), dev))
            side.stream.wait_event(ev_out)
            hip.gemm_rows(f1c, self._w("fc2", self.mlp.fc2, f32), bias=self.mlp.fc2.bias, residual=x2_out, row_scale=drop_m, out=out_out)
            side.done = side.stream.record_event()

    def _cls_chain(self, x_cls_in, o_c, B, T, drop_s, drop_m, x2_out, out_out):
        """x_cls_in: (B, D) fp32 CLS rows of the block input (a row-strided view is fine); o_c: (B*T, D) fp32 attention output of the CLS query of
        every frame (alpro_attn_fwd's cls_out).  Writes the CLS rows before the MLP into x2_out and the CLS rows of the block output into
        out_out (both (B, D), row-strided views of the token tensors are fine): three alpro_gemm_rows_f32 launches (norm2 fused into fc1's
        operand load) and alpro_cls_mean_residual for the frame mean -- no torch glue in between."""
        sa = self.attn
        f32 = torch.float32
        p_c = hip.gemm_rows(o_c, self._w("s_proj", sa.proj, f32), bias=sa.proj.bias, row_scale=drop_s)
        hip.cls_mean_residual(x_cls_in, p_c, x2_out, B, T)
        f1c = hip.gemm_rows(x2_out, self._w("fc1", self.mlp.fc1, f32), bias=self.mlp.fc1.bias, act=hip.ACT_GELU, ln=(self.norm2.weight, self.norm2.bias, VIT_EPS))
        hip.gemm_rows(f1c, self._w("fc2", self.mlp.fc2, f32), bias=self.mlp.fc2.bias, residual=x2_out, row_scale=drop_m, out=out_out)

    def _drop(self, rows, device):
        if not isinstance(self.drop_path, DropPath):
            return None
        pre = getattr(self, "_presampled", None)
        if pre is not None and rows in pre:      # sampled for all 12 blocks at once by sample_drop_paths (4 launches, not 132)
            return pre.pop(rows)
        return self.drop_path.row_scale(rows, device)

    def _forward_halves_unfused(self, x, xf, a, B, T, N, H, D, dt, drop_t=None, drop_s=None):
        """Round-2 form of the two attention halves' tails (fp32 residual read-modify-write in the GEMM epilogue, CLS side buffer);
        kept for A/B measurements (Block.fuse_residual_ln = False) and for the unmerged temporal projection."""
        ta, sa = self.temporal_attn, self.attn
        if self.merge_temporal_proj:
            mg = self._merged_tproj(dt)
            hip.gemm(a, mg["w"], out=xf, bias=mg["b1"], bias2=self.temporal_fc.bias, out_dtype=torch.float32, residual=xf,
                     row_scale=drop_t, row_scale_group=T, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
        else:
            pr = hip.gemm(a, self._w("t_proj", ta.proj, dt), bias=ta.proj.bias, row_scale=drop_t, row_scale_group=T)
            hip.gemm(pr, self._w("t_fc", self.temporal_fc, dt), out=xf, bias=self.temporal_fc.bias, out_dtype=torch.float32,
                     residual=xf, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
        hs = hip.layernorm(x, self.norm1.weight, self.norm1.bias, VIT_EPS, dt, rows=B * T * (N + 1),
                           map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N)
        qkv = hip.gemm(hs, self._w("s_qkv", sa.qkv, dt), bias=sa.qkv.bias)
        a = hip.attn(qkv, B * T, N + 1, H, sa.scale)
        side = torch.empty((B * T, D), dtype=torch.float32, device=x.device)
        hip.gemm(a, self._w("s_proj", sa.proj, dt), out=xf, bias=sa.proj.bias, out_dtype=torch.float32, residual=xf,
                 row_scale=drop_s, row_scale_group=N + 1,
                 map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N, side=side)
        hip.cls_mean_residual(x, side, x, B, T)

    def forward(self, x, B, T, W):
        """x: (B, 1 + N*T, D) fp32 contiguous token tensor; updated IN PLACE and returned (inference path)."""
        dt = rt.compute_dtype()
        S, D = x.shape[1], x.shape[2]
        N = (S - 1) // T
        H = self.attn.num_heads
        xf = x.view(B * S, D)
        ta, sa = self.temporal_attn, self.attn
        drop_t, drop_s, drop_m = self._drop(B * N, x.device), self._drop(B * T, x.device), self._drop(B, x.device)
        cp = rt.cls_precise(dt) and self.fuse_residual_ln and self.merge_temporal_proj
        side = _ClsSide.get(x.device) if (cp and rt.cls_stream(False) and x.is_cuda) else None
        if side is not None:
            x_cls_in, cls_q, o_c_buf = self._cls_side_begin(side, x, B, T, snapshot=True)
        else:
            x_cls_in = x[:, 0].clone() if cp else None   # a COPY (x is updated in place below; .contiguous() of a one-clip batch is a view); the temporal half never touches the CLS row
        # ---- temporal (vit.py:146-162)
        h = hip.layernorm(x, self.temporal_norm1.weight, self.temporal_norm1.bias, VIT_EPS, dt, rows=B * N * T,
                          map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
        if rt.fuse_temporal_attention() and hip.qkv_tattn_ok(h, T) and ta.qkv.bias is not None:
            # round 6: qkv Linear + frame attention in one launch, q | k | v consumed out of the accumulators (alpro_gemm_qkv_tattn) -- the
            # (B*N*T, 2304) tensor is never written.  Forward only: forward_train keeps the two launches, whose backward needs q, k, v
            a = hip.gemm_qkv_tattn(h, self._w("t_qkv", ta.qkv, dt), ta.qkv.bias, T, H, ta.scale)
        else:
            qkv = hip.gemm(h, self._w("t_qkv", ta.qkv, dt), bias=ta.qkv.bias)
            a = hip.attn_temporal(qkv, T, H, ta.scale)
        if self.fuse_residual_ln and self.merge_temporal_proj:
            # round 3: the two N = 768 projections write 16-bit deltas in plain row order; residual add + row maps + LayerNorm are one
            # streaming kernel each (alpro_add_layernorm_fwd) -- see the kernel's header comment in csrc/core.hip
            mg = self._merged_tproj(dt)
            d_t = hip.gemm(a, mg["w"], bias=mg["b1"], row_scale=drop_t, row_scale_group=T)
            if side is not None:
                side.wait_done()   # the previous block's chain has written this block's input CLS rows (first read: the kernel below)
            # round 6 (alpro_amd.config.defer_temporal_add): x + temporal branch (vit.py:162) is not written here -- the add + norm2 kernel below
            # adds both branches to the block input in the same order of fp32 additions (bit for bit the same x')
            defer = rt.defer_temporal_add()
            hs, _ = hip.add_layernorm(x, d_t, self.norm1.weight, self.norm1.bias, VIT_EPS, mode=hip.ADD_PRE_SPATIAL, x_out=None if defer else x,
                                      want_x=not defer, delta_bias=self.temporal_fc.bias, T=T, N=N)
            qkv = hip.gemm(hs, self._w("s_qkv", sa.qkv, dt), bias=sa.qkv.bias)
            if side is not None:
                torch.cuda.current_stream().wait_event(side.ev_q)
                a, o_c = hip.attn(qkv, B * T, N + 1, H, sa.scale, cls_q=cls_q, cls_group=T, cls_out=o_c_buf)
                ev_attn = torch.cuda.current_stream().record_event()
            elif cp:   # the CLS query of every frame once more in fp32, inside the same attention launch
                a, o_c = hip.attn(qkv, B * T, N + 1, H, sa.scale, cls_q=self._cls_qkv(x_cls_in), cls_group=T)
            else:
                a = hip.attn(qkv, B * T, N + 1, H, sa.scale)
            d_s = hip.gemm(a, self._w("s_proj", sa.proj, dt), bias=sa.proj.bias, row_scale=drop_s, row_scale_group=N + 1)
            if defer:
                h2 = hip.add_layernorm_pre_mlp2(x, d_t, self.temporal_fc.bias, d_s, self.norm2.weight, self.norm2.bias, VIT_EPS, T, N, x_out=x)
            else:
                h2, _ = hip.add_layernorm(x, d_s, self.norm2.weight, self.norm2.bias, VIT_EPS, mode=hip.ADD_PRE_MLP, x_out=x, T=T, N=N)
        else:
            self._forward_halves_unfused(x, xf, a, B, T, N, H, D, dt, drop_t, drop_s)
            h2 = hip.layernorm(x, self.norm2.weight, self.norm2.bias, VIT_EPS, dt)
        f1 = hip.gemm(h2, self._w("fc1", self.mlp.fc1, dt), bias=self.mlp.fc1.bias, act=hip.ACT_GELU)
        hip.gemm(f1, self._w("fc2", self.mlp.fc2, dt), out=xf, bias=self.mlp.fc2.bias, out_dtype=torch.float32, residual=xf,
                 row_scale=drop_m, row_scale_group=S)
        if side is not None:
            self._cls_side_finish(side, ev_attn, torch.cuda.current_stream().record_event(), x_cls_in, o_c, B, T, drop_s, drop_m,
                                  side.buf("x2_c", (B, D), x.device), x[:, 0])
        elif cp:
            self._cls_chain(x_cls_in, o_c, B, T, drop_s, drop_m, torch.empty_like(x_cls_in), x[:, 0])
        return x

    # ---- training path: fresh buffers (the backward needs every LayerNorm input), explicit backward ----------
    def forward_train(self, x, B, T, W):
        """Same arithmetic as forward() but out of place; returns (out, saved)."""
        dt = rt.compute_dtype()
        S, D = x.shape[1], x.shape[2]
        N = (S - 1) // T
        H = self.attn.num_heads
        ta, sa = self.temporal_attn, self.attn
        dev = x.device
        sv = {"x": x, "dims": (B, T, N, S, D, H), "dt": dt}
        sv["drop_t"], sv["drop_s"], sv["drop_m"] = self._drop(B * N, dev), self._drop(B * T, dev), self._drop(B, dev)
        cside = None
        if rt.cls_precise(dt) and rt.cls_stream(True) and x.is_cuda and self.fuse_residual_ln and self.merge_temporal_proj:
            cside = _ClsSide.get(dev)
            x_cls_in, cls_q, o_c_buf = self._cls_side_begin(cside, x, B, T, snapshot=False)
        h = hip.layernorm(x, self.temporal_norm1.weight, self.temporal_norm1.bias, VIT_EPS, dt, rows=B * N * T,
                          map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
        if rt.fuse_temporal_attention(training=True) and hip.qkv_tattn_ok(h, T) and ta.qkv.bias is not None:
            # round 6 (ALPRO_FUSE_TATTN=1; off by default: alpro_amd/config.py): one launch; q | k | v are still written (the backward reads them), what goes away is the attention launch and its re-read
            a_t, qkv_t, lse_t = hip.gemm_qkv_tattn(h, self._w("t_qkv", ta.qkv, dt), ta.qkv.bias, T, H, ta.scale, want_qkv=True)
        else:
            qkv_t = hip.gemm(h, self._w("t_qkv", ta.qkv, dt), bias=ta.qkv.bias)
            a_t, lse_t = hip.attn_temporal(qkv_t, T, H, ta.scale, want_lse=True)
        sv["merged"] = self.merge_temporal_proj
        if self.fuse_residual_ln and self.merge_temporal_proj:
            pr = None
            mg = self._merged_tproj(dt)
            d_t = hip.gemm(a_t, mg["w"], bias=mg["b1"], row_scale=sv["drop_t"], row_scale_group=T)
            if cside is not None:
                cside.wait_done()   # the previous block's chain has written this block's input CLS rows (first read: the kernel below)
            hs, xt = hip.add_layernorm(x, d_t, self.norm1.weight, self.norm1.bias, VIT_EPS, mode=hip.ADD_PRE_SPATIAL,
                                       delta_bias=self.temporal_fc.bias, T=T, N=N)
            del d_t
            qkv_s = hip.gemm(hs, self._w("s_qkv", sa.qkv, dt), bias=sa.qkv.bias)
            o_c = None
            if cside is not None:
                torch.cuda.current_stream().wait_event(cside.ev_q)
                a_s, lse_s, o_c = hip.attn(qkv_s, B * T, N + 1, H, sa.scale, want_lse=True, cls_q=cls_q, cls_group=T, cls_out=o_c_buf)
            elif rt.cls_precise(dt):
                a_s, lse_s, o_c = hip.attn(qkv_s, B * T, N + 1, H, sa.scale, want_lse=True, cls_q=self._cls_qkv(x[:, 0]), cls_group=T)
            else:
                a_s, lse_s = hip.attn(qkv_s, B * T, N + 1, H, sa.scale, want_lse=True)
            d_s = hip.gemm(a_s, self._w("s_proj", sa.proj, dt), bias=sa.proj.bias, row_scale=sv["drop_s"], row_scale_group=N + 1)
            h2, x2 = hip.add_layernorm(xt, d_s, self.norm2.weight, self.norm2.bias, VIT_EPS, mode=hip.ADD_PRE_MLP, T=T, N=N)
            del d_s
            if cside is not None:
                ev_x2 = torch.cuda.current_stream().record_event()   # o_c and the main path's x2 (whose CLS rows the chain overwrites) are written
        else:
            xt = torch.empty_like(x)
            xt[:, 0] = x[:, 0]
            if self.merge_temporal_proj:
                pr = None
                mg = self._merged_tproj(dt)
                hip.gemm(a_t, mg["w"], out=xt.view(B * S, D), bias=mg["b1"], bias2=self.temporal_fc.bias, out_dtype=torch.float32,
                         residual=x.view(B * S, D), row_scale=sv["drop_t"], row_scale_group=T, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
            else:
                pr = hip.gemm(a_t, self._w("t_proj", ta.proj, dt), bias=ta.proj.bias, row_scale=sv["drop_t"], row_scale_group=T)
                hip.gemm(pr, self._w("t_fc", self.temporal_fc, dt), out=xt.view(B * S, D), bias=self.temporal_fc.bias, out_dtype=torch.float32,
                         residual=x.view(B * S, D), map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
            hs = hip.layernorm(xt, self.norm1.weight, self.norm1.bias, VIT_EPS, dt, rows=B * T * (N + 1),
                               map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N)
            qkv_s = hip.gemm(hs, self._w("s_qkv", sa.qkv, dt), bias=sa.qkv.bias)
            a_s, lse_s = hip.attn(qkv_s, B * T, N + 1, H, sa.scale, want_lse=True)
            x2 = torch.empty_like(x)
            side = torch.empty((B * T, D), dtype=torch.float32, device=dev)
            hip.gemm(a_s, self._w("s_proj", sa.proj, dt), out=x2.view(B * S, D), bias=sa.proj.bias, out_dtype=torch.float32,
                     residual=xt.view(B * S, D), row_scale=sv["drop_s"], row_scale_group=N + 1,
                     map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N, side=side)
            hip.cls_mean_residual(xt, side, x2, B, T)
            h2 = hip.layernorm(x2, self.norm2.weight, self.norm2.bias, VIT_EPS, dt)
        u, sv["u_tiled"] = tr.gelu_save_buffer(B * S, self.mlp.fc1.out_features, D, dt, dev)
        sv["u_grad"] = tr.SAVE_GELU_GRAD   # u holds gelu'(fc1 output) (round 3) rather than the fc1 output itself (round 5: in the GEMM's tile order)
        f1 = hip.gemm(h2, self._w("fc1", self.mlp.fc1, dt), bias=self.mlp.fc1.bias, act=hip.ACT_GELU_SAVE_GRAD if sv["u_grad"] else hip.ACT_GELU, pre_act=u,
                      c2_tiled=sv["u_tiled"])
        out = torch.empty_like(x)
        hip.gemm(f1, self._w("fc2", self.mlp.fc2, dt), out=out.view(B * S, D), bias=self.mlp.fc2.bias, out_dtype=torch.float32,
                 residual=x2.view(B * S, D), row_scale=sv["drop_m"], row_scale_group=S)
        if cside is not None:
            self._cls_side_finish(cside, ev_x2, torch.cuda.current_stream().record_event(), x_cls_in, o_c, B, T, sv["drop_s"], sv["drop_m"], x2[:, 0], out[:, 0])
        elif rt.cls_precise(dt) and self.fuse_residual_ln and self.merge_temporal_proj:
            # precise CLS rows: the block output's CLS row and the saved pre-MLP stream's CLS row (norm2's backward input) take the fp32 values;
            # the backward differentiates the 16-bit graph as before (its CLS-row operands differ from these by one rounding)
            self._cls_chain(x[:, 0], o_c, B, T, sv["drop_s"], sv["drop_m"], x2[:, 0], out[:, 0])
        sv.update(h=h, qkv_t=qkv_t, a_t=a_t, lse_t=lse_t, pr=pr, xt=xt, hs=hs, qkv_s=qkv_s, a_s=a_s, lse_s=lse_s, x2=x2, h2=h2, u=u, f1=f1)
        return out, sv

    def forward_cls(self, x, B, T, W):
        """LAST block when only the CLS output is consumed (the frozen prompter: get_pseudo_labels uses feat = proj(x[:, 0])):
        the temporal half and the spatial K/V need every token, but the spatial projection, the MLP and what follows are
        evaluated on the CLS rows alone -- (B*T, D) / (B, D) instead of (B*S, D).  Same arithmetic as forward() for those rows
        (vit.py:165-212); eval mode only (no drop-path).  Returns the block output's CLS rows, (B, D) fp32."""
        dt = rt.compute_dtype()
        S, D = x.shape[1], x.shape[2]
        N = (S - 1) // T
        H = self.attn.num_heads
        xf = x.view(B * S, D)
        ta, sa = self.temporal_attn, self.attn
        h = hip.layernorm(x, self.temporal_norm1.weight, self.temporal_norm1.bias, VIT_EPS, dt, rows=B * N * T,
                          map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
        if rt.fuse_temporal_attention() and hip.qkv_tattn_ok(h, T) and ta.qkv.bias is not None:
            a = hip.gemm_qkv_tattn(h, self._w("t_qkv", ta.qkv, dt), ta.qkv.bias, T, H, ta.scale)
        else:
            qkv = hip.gemm(h, self._w("t_qkv", ta.qkv, dt), bias=ta.qkv.bias)
            a = hip.attn_temporal(qkv, T, H, ta.scale)
        if self.fuse_residual_ln and self.merge_temporal_proj:
            mg = self._merged_tproj(dt)
            d_t = hip.gemm(a, mg["w"], bias=mg["b1"])
            hs, _ = hip.add_layernorm(x, d_t, self.norm1.weight, self.norm1.bias, VIT_EPS, mode=hip.ADD_PRE_SPATIAL, want_x=False,
                                      delta_bias=self.temporal_fc.bias, T=T, N=N)   # the patch rows of x are not read again; x[:, 0] is untouched
        else:
            if self.merge_temporal_proj:
                mg = self._merged_tproj(dt)
                hip.gemm(a, mg["w"], out=xf, bias=mg["b1"], bias2=self.temporal_fc.bias, out_dtype=torch.float32, residual=xf,
                         map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
            else:
                pr = hip.gemm(a, self._w("t_proj", ta.proj, dt), bias=ta.proj.bias)
                hip.gemm(pr, self._w("t_fc", self.temporal_fc, dt), out=xf, bias=self.temporal_fc.bias, out_dtype=torch.float32,
                         residual=xf, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
            hs = hip.layernorm(x, self.norm1.weight, self.norm1.bias, VIT_EPS, dt, rows=B * T * (N + 1),
                               map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N)
        qkv = hip.gemm(hs, self._w("s_qkv", sa.qkv, dt), bias=sa.qkv.bias)
        a = hip.attn(qkv, B * T, N + 1, H, sa.scale)
        a_cls = a.view(B * T, N + 1, D)[:, 0].contiguous()                                  # CLS query of every frame
        p_cls = hip.gemm(a_cls, self._w("s_proj", sa.proj, dt), bias=sa.proj.bias, out_dtype=torch.float32)
        x_cls = x[:, 0] + p_cls.view(B, T, D).mean(1)                                        # frame mean of the CLS rows (vit.py:187)
        h2 = hip.layernorm(x_cls.contiguous(), self.norm2.weight, self.norm2.bias, VIT_EPS, dt)
        f1 = hip.gemm(h2, self._w("fc1", self.mlp.fc1, dt), bias=self.mlp.fc1.bias, act=hip.ACT_GELU)
        return hip.gemm(f1, self._w("fc2", self.mlp.fc2, dt), bias=self.mlp.fc2.bias, out_dtype=torch.float32, residual=x_cls.contiguous())

    def _wt(self, name, lin, dt):
        return tr.transposed_operand(self._ops, name + "^T", lin.weight, dt)

    def _merged_tproj_backward(self, sv, dx, B, T, N, D, dt, G=None):
        """Backward of xt[:, 1:] = x[:, 1:] + s * (a We^T + Wfc bp) + bfc (see merge_temporal_proj); returns d(a).
        G: s * d(xt)[:, 1:] in the operand dtype when norm1's backward already emitted it (with dbfc accumulated), else built here."""
        ta, fc = self.temporal_attn, self.temporal_fc
        wp, bp, wf = ta.proj.weight.detach(), ta.proj.bias.detach(), fc.weight.detach()
        mg = self._merged_tproj(dt)
        dev = dx.device
        # G = s * dY in the operand dtype; dbfc = colsum(dY) (unscaled) from the same pass
        if G is None:
            G = hip.gather_cast(dx, dt, rows=B * N * T, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T, row_scale=sv["drop_t"], row_scale_group=T,
                                colsum_pre=tr.bias_grad(fc.bias))
        ws = sv.get("ws")  # zeroed slice of the per-backward workspace (one fill for all 12 blocks), else allocate
        if ws is None:
            ws = torch.zeros(D * D + D, dtype=torch.float32, device=dev)
        dWe, db1 = ws[:D * D].view(D, D), ws[D * D:]
        if dt != torch.float32:
            hip.gemm_tn_acc(G, sv["a_t"], dWe, colsum=db1)                       # dWe = G^T a, db1 = colsum(G)
        else:
            hip.gemm(hip.transpose(G, colsum=db1), hip.transpose(sv["a_t"]), out=dWe, out_dtype=torch.float32)
        da = tr.dgrad(G, mg["wT"])                                              # d(a) = G We
        if sv.get("defer_product_rule"):   # the bank applies the product rule for a whole group of blocks at once (_MergedTProjBank.product_rule)
            return da
        # product rule back onto the two real parameters (768^3 each).  Exact mode: fp32 MFMA.  16-bit modes: 16-bit operands with fp32
        # accumulation into the fp32 gradient, like every other weight gradient on this path (dWe itself came from 16-bit operands) --
        # the two fp32 768^3 GEMMs were 57 us each on 36 workgroups, 2 x 12 of them per step.
        g_fc, g_p = tr.grad_buffer(fc.weight, zero=True)[0], tr.grad_buffer(ta.proj.weight, zero=True)[0]
        if dt != torch.float32:
            hip.gemm(hip.cast(dWe, dt), self._w("t_proj", ta.proj, dt), out=g_fc, out_dtype=torch.float32, residual=g_fc)   # += dWe Wp^T
            hip.gemm(self._wt("t_fc", fc, dt)[:, :D], hip.transpose(dWe, out_dtype=dt), out=g_p, out_dtype=torch.float32, residual=g_p)  # += Wfc^T dWe
        else:
            hip.gemm(dWe, wp.contiguous(), out=g_fc, out_dtype=torch.float32, residual=g_fc)                     # += dWe Wp^T
            hip.gemm(hip.transpose(wf.contiguous()), hip.transpose(dWe), out=g_p, out_dtype=torch.float32, residual=g_p)  # += Wfc^T dWe
        g_fc.addr_(db1, bp)                                                                                       # += db1 bp^T
        tr.bias_grad(ta.proj.bias).add_(torch.mv(wf.t(), db1))                                                    # += Wfc^T db1
        return da

    def backward(self, sv, dx, dz=None, emit_for=None):
        """dx: gradient w.r.t. the block output, (B, S, D) fp32; overwritten with the gradient w.r.t. the block input.
        dz: the operand rows drop_m * dx in the compute dtype if whoever produced dx already emitted them (the final norm's backward, or
        the next block's temporal-LayerNorm backward), else None -> built here by alpro_gather_cast.
        emit_for: saved state of the PREVIOUS block (the next one to run its backward): its dz is emitted by this block's last kernel.
        Returns (dx, dz for the previous block or None)."""
        B, T, N, S, D, H = sv["dims"]
        dt = sv["dt"]
        ta, sa = self.temporal_attn, self.attn
        fuse = self.fuse_ln_bwd_emit
        # ---- MLP: out = x2 + drop_m * (fc2(gelu(fc1(LN2(x2)))))
        if dz is None:
            dz = hip.gather_cast(dx, dt, row_scale=sv["drop_m"], row_scale_group=S)
        tr.wgrad(dz, sv["f1"], self.mlp.fc2.weight, self.mlp.fc2.bias)
        du = tr.dgrad(dz, self._wt("fc2", self.mlp.fc2, dt), gelu_pre=sv["u"], gelu_saved_grad=sv.get("u_grad", False), gelu_tiled=sv.get("u_tiled", False))
        del dz
        tr.wgrad(du, sv["h2"], self.mlp.fc1.weight, self.mlp.fc1.bias)
        dh2 = tr.dgrad(du, self._wt("fc1", self.mlp.fc1, dt))
        g, b_ = tr.grad_buffer(self.norm2.weight, zero=True)[0], tr.grad_buffer(self.norm2.bias, zero=True)[0]
        # ---- spatial: x2 = scatter(xt + drop_s * proj(attn(qkv(LN1(gather(xt)))))), CLS averaged over frames
        if fuse:   # norm2's backward hands the finished d(x2) rows straight to the spatial projection's GEMMs (frame-token order, drop_s, CLS / T)
            _, dpo = hip.layernorm_bwd(dh2, sv["x2"], self.norm2.weight, VIT_EPS, dx, g, b_,
                                       emit=dict(mode=hip.EMIT_FRAME, rows=B * T * (N + 1), dtype=dt, T=T, N=N, scale=sv["drop_s"]))
        else:
            hip.layernorm_bwd(dh2, sv["x2"], self.norm2.weight, VIT_EPS, dx, g, b_)
            dpo = hip.gather_cast(dx, dt, rows=B * T * (N + 1), map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N,
                                  row_scale=sv["drop_s"], row_scale_group=N + 1, cls_scale=1.0 / T)
        tr.wgrad(dpo, sv["a_s"], sa.proj.weight, sa.proj.bias)
        da = tr.dgrad(dpo, self._wt("s_proj", sa.proj, dt))
        del dpo
        dqkv = hip.attn_bwd(sv["qkv_s"], sv["a_s"], da, sv["lse_s"], B * T, N + 1, H, sa.scale)
        tr.wgrad(dqkv, sv["hs"], sa.qkv.weight, sa.qkv.bias)
        dhs = tr.dgrad(dqkv, self._wt("s_qkv", sa.qkv, dt))
        g, b_ = tr.grad_buffer(self.norm1.weight, zero=True)[0], tr.grad_buffer(self.norm1.bias, zero=True)[0]
        # ---- temporal: xt[:, 1:] = x[:, 1:] + fc(drop_t * proj(attn(qkv(LN_t(x[:, 1:])))))
        G = None
        if fuse and sv["merged"]:  # norm1's backward emits drop_t * d(xt)[:, 1:] and the temporal_fc bias gradient (unscaled column sums)
            _, G = hip.layernorm_bwd(dhs, sv["xt"], self.norm1.weight, VIT_EPS, dx, g, b_, rows=B * T * (N + 1), map_mode=hip.MAP_FRAME_TOKENS,
                                     map_p0=T, map_p1=N, emit=dict(mode=hip.EMIT_SKIP_CLS, rows=B * N * T, dtype=dt, T=T, N=N, scale=sv["drop_t"], group=T,
                                                                   colsum_pre=tr.bias_grad(self.temporal_fc.bias)))
        else:
            hip.layernorm_bwd(dhs, sv["xt"], self.norm1.weight, VIT_EPS, dx, g, b_, rows=B * T * (N + 1),
                              map_mode=hip.MAP_FRAME_TOKENS, map_p0=T, map_p1=N)
        if sv["merged"]:
            da = self._merged_tproj_backward(sv, dx, B, T, N, D, dt, G=G)
        else:
            dfo = hip.gather_cast(dx, dt, rows=B * N * T, map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
            tr.wgrad(dfo, sv["pr"], self.temporal_fc.weight, self.temporal_fc.bias)
            dpp = tr.dgrad(dfo, self._wt("t_fc", self.temporal_fc, dt), row_scale=sv["drop_t"], row_scale_group=T)
            tr.wgrad(dpp, sv["a_t"], ta.proj.weight, ta.proj.bias)
            da = tr.dgrad(dpp, self._wt("t_proj", ta.proj, dt))
        dqkv = hip.attn_temporal_bwd(sv["qkv_t"], sv["a_t"], da, sv["lse_t"], T, H, ta.scale)
        tr.wgrad(dqkv, sv["h"], ta.qkv.weight, ta.qkv.bias)
        dh = tr.dgrad(dqkv, self._wt("t_qkv", ta.qkv, dt))
        g, b_ = tr.grad_buffer(self.temporal_norm1.weight, zero=True)[0], tr.grad_buffer(self.temporal_norm1.bias, zero=True)[0]
        dz_prev = None
        if fuse and emit_for is not None:  # ... and the temporal norm's backward emits the previous block's MLP operand (its drop_m, all rows incl. CLS)
            _, dz_prev = hip.layernorm_bwd(dh, sv["x"], self.temporal_norm1.weight, VIT_EPS, dx, g, b_, rows=B * N * T, map_mode=hip.MAP_SKIP_CLS,
                                           map_p0=N * T, emit=dict(mode=hip.EMIT_ROWS, rows=B * S, dtype=dt, T=T, N=N, scale=emit_for["drop_m"], group=S,
                                                                   extra_cls=B))
        else:
            hip.layernorm_bwd(dh, sv["x"], self.temporal_norm1.weight, VIT_EPS, dx, g, b_, rows=B * N * T,
                              map_mode=hip.MAP_SKIP_CLS, map_p0=N * T)
        return dx, dz_prev


_KEEP_CACHE = {}


def sample_drop_paths(blocks, B, T, N, device):
    """All stochastic-depth row masks of one training forward in ONE draw: per block three Bernoulli(keep_i) vectors of
    B*N (temporal), B*T (spatial) and B (MLP) rows, scaled by 1/keep_i (vit_utils.py:137-151, rates vit.py:272).  The reference
    draws them one by one (36 x {rand, add, floor, div}); the distribution is the same, the launch count is 4."""
    sizes = (B * N, B * T, B)
    live = [blk for blk in blocks if isinstance(blk.drop_path, DropPath) and blk.drop_path.drop_prob and blk.training]
    if not live:
        return
    per = sum(sizes)
    key = (per, str(device), tuple(blk.drop_path.drop_prob for blk in live))
    keep = _KEEP_CACHE.get(key)
    if keep is None:  # per-row keep probability, built once per (batch geometry, rates)
        keep = torch.tensor([1.0 - blk.drop_path.drop_prob for blk in live], dtype=torch.float32).repeat_interleave(per).to(device)
        _KEEP_CACHE.clear()
        _KEEP_CACHE[key] = keep
    scale = torch.floor(keep + torch.rand(per * len(live), dtype=torch.float32, device=device)) / keep
    off = 0
    for blk in live:
        blk._presampled = {}
        for n in sizes:
            blk._presampled[n] = scale[off:off + n]
            off += n


class _MergedTProjBank:
    """Batched upkeep of the merged temporal projections of ALL blocks of one encoder (round 3).

    Per block and optimizer step the merged form needs W_e = W_fc W_p (fp32 768^3 product), its 16-bit copy, its transpose and
    W_fc b_p -- five small launches of which the product alone takes 64 us on 36 workgroups -- and the backward needs the product rule
    back onto the two real weights (two more 768^3 products, a cast, a transpose, a rank-1 update and a gemv): ~230 launches and 2.8 ms
    per training step for 12 blocks.  Here the same arithmetic runs as five launches for the forward refresh (alpro_transpose_batch,
    alpro_gemm_batch, one cast, alpro_transpose_batch, alpro_tproj_small) and five per group of blocks for the product rule, on buffers and
    job tables that live across steps.  Blocks outside a VisionTransformer (tests build single Blocks) keep the per-block code."""

    def __init__(self, blocks):
        self.blocks = list(blocks)
        self.state = None          # forward side: dict(key, dt, bufs, tables)
        self.bwd = None            # backward side: persistent ws / 16-bit copies
        self._bwd_tables = {}

    def __reduce__(self):
        """copy.deepcopy / pickle of a model must not drag the job tables (ctypes pointers into THIS model's storage) along: the copy starts
        with an empty bank and VisionTransformer._attach_bank() rebuilds it over the copied blocks at the next forward."""
        return (_MergedTProjBank, ([],))

    def _params(self):
        return [p for b in self.blocks for p in (b.temporal_fc.weight, b.temporal_attn.proj.weight, b.temporal_attn.proj.bias)]

    def get(self, idx, dt):
        """-> dict(w = W_e (D, D) in dt, wT = W_e^T in dt, b1 = W_fc b_p fp32) of block idx, refreshing ALL blocks if any parameter changed."""
        from alpro_amd.modeling import weights
        ps = self._params()
        fast = (weights.param_epoch(), weights._EXT_EPOCH[0], dt, tuple(p._version for p in ps))
        st = self.state
        if st is None or st["fast"] != fast:
            key = (dt,) + tuple(param_version(p) for p in ps)   # frozen parameters (the prompter) keep their key across optimizer steps
            if st is None or st["key"] != key:
                st = self._refresh(dt, key)
            st["fast"] = fast
        return dict(w=st["we_dt"][idx], wT=st["weT_dt"][idx], b1=st["b1"][idx])

    def _refresh(self, dt, key):
        nb = len(self.blocks)
        wf = [b.temporal_fc.weight.detach() for b in self.blocks]
        wp = [b.temporal_attn.proj.weight.detach() for b in self.blocks]
        bp = [b.temporal_attn.proj.bias.detach() for b in self.blocks]
        D, dev = wf[0].shape[0], wf[0].device
        st = self.state
        if st is None or st["dt"] != dt or st["we32"].device != dev:
            we32 = torch.empty((nb, D, D), dtype=torch.float32, device=dev)
            st = dict(dt=dt, we32=we32, wpT32=torch.empty_like(we32), b1=torch.empty((nb, D), dtype=torch.float32, device=dev),
                      we_dt=we32 if dt == torch.float32 else torch.empty((nb, D, D), dtype=dt, device=dev),
                      weT_dt=torch.empty((nb, D, D), dtype=dt, device=dev), sig=None)
        sig = tuple(t.data_ptr() for t in wf + wp + bp)
        if st["sig"] != sig:   # the parameters moved (FlatAdamW adopted them into its flat buffer): rebuild the job tables
            st["t_wp"] = hip.transpose_jobs([(wp[i], st["wpT32"][i]) for i in range(nb)])
            gb = hip.GemmBatch()
            for i in range(nb):
                gb.add(wf[i], st["wpT32"][i], out=st["we32"][i], out_dtype=torch.float32)   # W_e = W_fc W_p
            st["gemm"] = gb
            st["t_we"] = hip.transpose_jobs([(st["we32"][i], st["weT_dt"][i]) for i in range(nb)])
            st["t_b1"] = hip.tproj_jobs([dict(wfc=wf[i], bp=bp[i], b1=st["b1"][i]) for i in range(nb)], dev)
            st["sig"] = sig
        with torch.no_grad():
            hip.transpose_batch(*st["t_wp"], torch.float32)
            st["gemm"].launch()
            if dt != torch.float32:
                hip.cast(st["we32"].view(-1), dt, out=st["we_dt"].view(-1))
            hip.transpose_batch(*st["t_we"], dt)
            hip.tproj_small(*st["t_b1"], D, 0)
        st["key"] = key
        self.state = st
        from alpro_amd.modeling import weights
        weights.note_operand_rebuild()
        return st

    # ---- backward ------------------------------------------------------------------------------------------------
    def workspace(self, dt, device):
        """(nb, D*D + D) fp32, zeroed: row i receives dW_e (D, D) and db1 (D) of block i from its weight-gradient GEMM."""
        nb, D = len(self.blocks), self.blocks[0].temporal_fc.weight.shape[0]
        b = self.bwd
        if b is None or b["ws"].device != device or b["dt"] != dt:
            ws = torch.empty((nb, D * D + D), dtype=torch.float32, device=device)
            b = self.bwd = dict(dt=dt, ws=ws, ws_dt=torch.empty((nb, D * D + D), dtype=dt, device=device) if dt != torch.float32 else None,
                                dweT=torch.empty((nb, D, D), dtype=dt, device=device))
            self._bwd_tables = {}
        b["ws"].zero_()
        return b["ws"]

    def product_rule(self, lo, hi, dt):
        """Blocks lo..hi-1: dW_fc += dW_e W_p^T + db1 b_p^T, dW_p += W_fc^T dW_e, db_p += W_fc^T db1 -- 16-bit operands with fp32 accumulation
        into the fp32 gradients, like every other weight gradient on this path.  Five launches for the whole group."""
        assert dt != torch.float32 and self.bwd is not None
        b, blocks = self.bwd, self.blocks[lo:hi]
        D = blocks[0].temporal_fc.weight.shape[0]
        ops = []
        for i, blk in zip(range(lo, hi), blocks):
            ta, fc = blk.temporal_attn, blk.temporal_fc
            g_fc, g_p = tr.grad_buffer(fc.weight, zero=True)[0], tr.grad_buffer(ta.proj.weight, zero=True)[0]
            g_bp = tr.bias_grad(ta.proj.bias)
            ops.append(dict(i=i, wp_dt=blk._w("t_proj", ta.proj, dt), wfT_dt=blk._wt("t_fc", fc, dt), g_fc=g_fc, g_p=g_p, g_bp=g_bp,
                            wf=fc.weight.detach(), bp=ta.proj.bias.detach()))
        sig = tuple((o["wp_dt"].data_ptr(), o["wfT_dt"].data_ptr(), o["g_fc"].data_ptr(), o["g_p"].data_ptr(), o["g_bp"].data_ptr(), o["wf"].data_ptr())
                    for o in ops)
        tb = self._bwd_tables.get((lo, hi))
        if tb is None or tb["sig"] != sig:
            ga, gbb = hip.GemmBatch(), hip.GemmBatch()
            for o in ops:
                dwe_dt = b["ws_dt"][o["i"], :D * D].view(D, D)
                ga.add(dwe_dt, o["wp_dt"], out=o["g_fc"], out_dtype=torch.float32, residual=o["g_fc"])                        # += dW_e W_p^T
                gbb.add(o["wfT_dt"][:, :D], b["dweT"][o["i"]], out=o["g_p"], out_dtype=torch.float32, residual=o["g_p"])     # += W_fc^T dW_e
            tb = dict(sig=sig, ga=ga, gb=gbb, keep=ops,
                      t_dwe=hip.transpose_jobs([(b["ws"][o["i"], :D * D].view(D, D), b["dweT"][o["i"]]) for o in ops]),
                      t_small=hip.tproj_jobs([dict(wfc=o["wf"], bp=o["bp"], db1=b["ws"][o["i"], D * D:], g_fc=o["g_fc"], g_bp=o["g_bp"]) for o in ops],
                                             b["ws"].device))
            self._bwd_tables[(lo, hi)] = tb
        hip.cast(b["ws"][lo:hi].reshape(-1), dt, out=b["ws_dt"][lo:hi].reshape(-1))
        hip.transpose_batch(*tb["t_dwe"], dt)
        tb["ga"].launch()
        tb["gb"].launch()
        hip.tproj_small(*tb["t_small"], D, 1)


class PatchEmbed(nn.Module):
    """Image to patch embedding: the stride-16 conv (vit.py:230) evaluated as im2col rows x weight GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        assert patch_size == (16, 16), "alpro_patchify is specialised for 16x16 patches (ViT-B/16)"
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 hybrid_backbone=None, norm_layer=nn.LayerNorm, num_frames=8, attention_type='divided_space_time', dropout=0.,
                 cross_attention_config=None, use_grad_checkpointing=False):
        super().__init__()
        assert drop_rate == 0., "drop_rate is 0 in every release config (pos_drop/time_drop are identities)"
        self.attention_type = attention_type
        self.depth = depth
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.time_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, self.depth)]  # vit.py:272
        self.blocks = nn.ModuleList([
            Block(layer_num=i, use_grad_checkpointing=use_grad_checkpointing, dim=embed_dim, num_heads=num_heads,
                  mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                  drop_path=dpr[i], norm_layer=norm_layer, attention_type=self.attention_type) for i in range(self.depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        for i, blk in enumerate(self.blocks):  # vit.py:290-298
            if i > 0:
                nn.init.constant_(blk.temporal_fc.weight, 0)
                nn.init.constant_(blk.temporal_fc.bias, 0)
        self._ops = OperandCache()
        self._attach_bank()

    def _attach_bank(self):
        """(Re)build the merged-projection bank over THIS module's blocks (also after copy.deepcopy, whose blocks come back with an empty bank
        and fall back to the per-block path until then)."""
        self._tproj_bank = _MergedTProjBank(self.blocks)    # (plain attribute: not a module, nothing for state_dict)
        for i, blk in enumerate(self.blocks):
            blk._bank_obj, blk._bank_idx = self._tproj_bank, i   # (a plain reference: copies / pickles reduce the bank to an empty one)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'time_embed'}

    def _embed(self, x):
        """(B, C, T, H, W) -> (B, 1 + N*T, D) fp32 tokens (vit.py:321-361)."""
        if not self._tproj_bank.blocks or self._tproj_bank.blocks[0] is not self.blocks[0]:
            self._attach_bank()
        dt = rt.compute_dtype()
        B, C, T, Hh, Ww = x.shape
        D = self.embed_dim
        N = (Hh // 16) * (Ww // 16)
        pidx, tidx = self._embed_index(N, T, Ww // 16)
        frames = x.transpose(1, 2).reshape(B * T, C, Hh, Ww).contiguous().float()
        rows = hip.patchify(frames, dt)
        w = self._ops.get("patch_w", self.patch_embed.proj.weight.view(D, -1), dt)
        # table[n*T + t] = conv bias + pos_embed[1 + n] + time_embed[t]
        pos_p, time_e = self.pos_embed.detach()[0, 1:], self.time_embed.detach()[0]
        if pidx is not None:
            pos_p = pos_p.index_select(0, pidx)
        if tidx is not None:
            time_e = time_e.index_select(0, tidx)
        table = (self.patch_embed.proj.bias.detach()[None, None, :] + pos_p[:, None, :] + time_e[None, :, :]).reshape(N * T, D).contiguous()
        tok = torch.empty((B, 1 + N * T, D), dtype=torch.float32, device=x.device)
        tok[:, 0] = (self.cls_token.detach() + self.pos_embed.detach()[:, :1]).view(1, D)
        hip.gemm(rows, w, out=tok.view(-1, D), out_dtype=torch.float32, residual=table, map_mode=hip.MAP_PATCH_EMBED, map_p0=T, map_p1=N)
        self._last_rows = rows
        return tok, T, Ww // 16, N

    def _embed_index(self, N, T, Wg):
        """Nearest-neighbour resampling of pos_embed / time_embed when the input grid differs from the tables
        (vit.py:328-340: the P x P patch table is resampled in 2-D to the (N / Wg) x Wg input grid; :350-357: the frame
        table in 1-D).  Returns (patch index (N,) or None, frame index (T,) or None)."""
        from alpro_amd.utils.load_save import nearest_index
        dev = self.pos_embed.device
        pidx = tidx = None
        P2 = self.pos_embed.size(1) - 1
        if N != P2:
            P = int(P2 ** 0.5)
            if P * P != P2 or N % Wg != 0:
                raise RuntimeError("pos_embed holds %d patch slots (not a square grid) / input has %d patches in rows of %d" % (P2, N, Wg))
            Hg = N // Wg
            pidx = (nearest_index(P, Hg, dev)[:, None] * P + nearest_index(P, Wg, dev)[None, :]).reshape(-1)
        if T != self.time_embed.size(1):
            tidx = nearest_index(self.time_embed.size(1), T, dev)
        return pidx, tidx

    def _embed_backward(self, rows, dtok, B, T, N, Wg=None):
        """Gradients of patch_embed.proj / cls_token / pos_embed / time_embed from dtok (B, 1+N*T, D)."""
        dt = rows.dtype
        D = self.embed_dim
        drows = hip.gather_cast(dtok, dt, rows=B * T * N, map_mode=hip.MAP_PATCH_EMBED, map_p0=T, map_p1=N)
        pe = self.patch_embed.proj
        if dt != torch.float32:  # 16-bit operands: in-place TN weight gradient (contraction over the B*T*N patch rows)
            gw = tr.grad_buffer(pe.weight, zero=True)[0]
            hip.gemm_tn_acc(drows, rows, gw.view(D, -1))
        else:
            gw, existed = tr.grad_buffer(pe.weight)
            gw2 = gw.view(D, -1)
            hip.gemm(hip.transpose(drows), hip.transpose(rows), out=gw2, out_dtype=torch.float32, residual=gw2 if existed else None)
        dtable = dtok[:, 1:].sum(0).view(N, T, D)  # small (N*T, D) reductions: parameter-sized, stay in torch
        tr.add_grad(pe.bias, dtable.sum((0, 1)))
        dcls = dtok[:, 0].sum(0)
        tr.add_grad(self.cls_token, dcls)
        pidx, tidx = self._embed_index(N, T, Wg) if Wg is not None else (None, None)
        dpos, dtime = dtable.sum(1), dtable.sum(0)
        if pidx is not None:  # resampled tables: scatter the gradients back onto the slots they were read from
            dpos = torch.zeros((self.pos_embed.size(1) - 1, D), dtype=dpos.dtype, device=dpos.device).index_put_((pidx,), dpos, accumulate=True)   # (sorted accumulate: fixed order, unlike index_add_'s atomics)
        if tidx is not None:
            dtime = torch.zeros((self.time_embed.size(1), D), dtype=dtime.dtype, device=dtime.device).index_put_((tidx,), dtime, accumulate=True)
        tr.add_grad(self.pos_embed, torch.cat([dcls[None], dpos], 0))
        tr.add_grad(self.time_embed, dtime)

    def forward_features(self, x, return_all_tokens=False):
        B = x.shape[0]
        tok, T, W, N = self._embed(x)
        tok = run_blocks(self.blocks, tok, B, T, W)
        y = hip.layernorm(tok, self.norm.weight, self.norm.bias, VIT_EPS, torch.float32).view(B, -1, self.embed_dim)
        return y if return_all_tokens else y[:, 0]

    def forward(self, x):
        raise NotImplementedError("the Kinetics classification head is outside ALPRO's video-text path")


default_cfgs = {'vit_base_patch16_224': {'num_classes': 1000, 'input_size': (3, 224, 224), 'first_conv': 'patch_embed.proj', 'classifier': 'head'}}


class TimeSformer(nn.Module):
    """Same constructor / forward_features contract as vit.py:419-503."""

    def __init__(self, model_cfg, input_format='BGR', cross_attention_config=None, **kwargs):
        super().__init__()
        self.config_file = str(model_cfg)
        self.img_size = model_cfg['img_size']
        self.patch_size = model_cfg['patch_size']
        self.num_frames = model_cfg['num_frm']
        self.attn_drop_rate = model_cfg['attn_drop_rate']
        self.drop_path_rate = model_cfg['drop_path_rate']
        self.drop_rate = model_cfg['drop_rate']
        self.use_pooling = model_cfg['use_maxpooling']
        self.use_grad_ckpt = model_cfg['gradient_checkpointing']
        if self.use_grad_ckpt:
            # config_release/timesformer_divst_8x32_224_k600_gc.json:9.  The reference re-runs each Block in backward to fit 16-40 GB devices
            # (vit.py:366-370); this build keeps every activation a step needs (64 GB at B = 64 x 8 frames of the 288 GB) and never recomputes:
            # the flag is accepted for config compatibility and has NO effect -- said once, not silently
            import warnings
            warnings.warn("alpro_amd: gradient_checkpointing=true is accepted and ignored -- activations are kept in HBM (288 GB per MI355X; "
                          "B=64 x 8 frames peaks at 64 GB), no block is recomputed in backward", RuntimeWarning, stacklevel=2)
        self.attention_type = 'divided_space_time'
        self.num_classes = 400
        self.input_format = input_format
        assert input_format == "RGB", "Official TimeSformer uses RGB input."
        self.model = VisionTransformer(img_size=self.img_size, num_classes=self.num_classes, patch_size=self.patch_size,
                                       embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                                       norm_layer=partial(nn.LayerNorm, eps=VIT_EPS), drop_rate=self.drop_rate,
                                       attn_drop_rate=self.attn_drop_rate, drop_path_rate=self.drop_path_rate,
                                       num_frames=self.num_frames, attention_type=self.attention_type,
                                       cross_attention_config=cross_attention_config, use_grad_checkpointing=self.use_grad_ckpt, **kwargs)
        if self.use_pooling:
            self.maxpool_kernel_size = model_cfg['maxpool_kernel_size']
            self.maxpooling = torch.nn.MaxPool2d(kernel_size=self.maxpool_kernel_size)
        self.model.default_cfg = default_cfgs['vit_base_patch' + str(self.patch_size) + '_224']
        self.num_patches = (self.img_size // self.patch_size) * (self.img_size // self.patch_size)

    def forward_features(self, x, return_all_tokens=True, pooling='temporal'):
        """x: (b, c, t, h, w) -> (b, 1 + h*w/256, 768): final LayerNorm fused with the temporal mean pool.
        With autograd enabled the whole encoder is one autograd node with a hand-written backward."""
        assert pooling == 'temporal' and return_all_tokens, "ALPRO only calls forward_features(return_all_tokens=True) with temporal pooling"
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return tr.run_anchored(_VisualRun(self), [x], list(self.parameters()))
        m = self.model
        B = x.shape[0]
        tok, T, W, N = m._embed(x)
        tok = run_blocks(m.blocks, tok, B, T, W)
        out32, _ = hip.vit_final_pool(tok, m.norm.weight, m.norm.bias, VIT_EPS, B, T, N, torch.float32)
        return out32

    def forward_cls(self, x):
        """x: (b, c, t, h, w) -> (b, 768): the CLS row of forward_features(x) only, inference only.  The last block's spatial
        projection and MLP run on the CLS rows alone (Block.forward_cls); used by the frozen prompter, whose pseudo labels
        depend on nothing else (alpro_models.py:531-551)."""
        assert not torch.is_grad_enabled(), "forward_cls is an inference path"
        m = self.model
        B = x.shape[0]
        tok, T, W, N = m._embed(x)
        tok = run_blocks(m.blocks[:-1], tok, B, T, W)
        cls = m.blocks[-1].forward_cls(tok, B, T, W)
        return hip.layernorm(cls, m.norm.weight, m.norm.bias, VIT_EPS, torch.float32)

    def forward(self, x):
        return self.model(x)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """A string is a pre-trained checkpoint source exactly as in the reference (vit.py:515-533: 'vit_base_patch16_224', a CLIP
        ViT file, or a Kinetics TimeSformer checkpoint path; remapped by timesformer/helpers.py); a mapping is a plain state_dict."""
        if isinstance(state_dict, str):
            from alpro_amd.modeling.timesformer.helpers import load_visual_checkpoint
            return load_visual_checkpoint(self, state_dict)
        return super().load_state_dict(state_dict, strict=strict, **kw)


class _VisualRun:
    """Forward/backward of the whole visual encoder for tr.Anchor (alpro_models.py:186-194 under autograd)."""

    def __init__(self, enc):
        self.enc = enc

    def forward(self, x):
        m = self.enc.model
        B = x.shape[0]
        tok, T, W, N = m._embed(x)
        self.rows, self.dims, self.Wg = m._last_rows, (B, T, N), W
        self.saved = []
        if B * N != B * T and B * T != B:  # (the table is keyed by row count: the three counts must differ, else draw per call)
            sample_drop_paths(m.blocks, B, T, N, tok.device)
        for blk in m.blocks:
            tok, sv = blk.forward_train(tok, B, T, W)
            blk._presampled = None
            self.saved.append(sv)
        _ClsSide.join(tok.device)
        self.tok = tok
        out32, _ = hip.vit_final_pool(tok, m.norm.weight, m.norm.bias, VIT_EPS, B, T, N, torch.float32)
        return out32

    def backward(self, dout):
        m = self.enc.model
        B, T, N = self.dims
        D = m.embed_dim
        # final norm + temporal mean pool (vit.py:372,484-492): every frame token receives dout / T, the CLS token dout
        dy = torch.empty((B, 1 + N * T, D), dtype=torch.float32, device=dout.device)
        dy[:, 0] = dout[:, 0]
        torch.mul(dout[:, 1:].unsqueeze(2).expand(B, N, T, D), 1.0 / T, out=dy[:, 1:].view(B, N, T, D))  # one broadcast pass (was mul + repeat_interleave + copy)
        dtok = torch.empty_like(dy)
        g, b_ = tr.grad_buffer(m.norm.weight, zero=True)[0], tr.grad_buffer(m.norm.bias, zero=True)[0]
        dz = None
        if m.blocks[-1].fuse_ln_bwd_emit:   # the final norm's backward emits the last block's MLP operand rows (its drop-path scale)
            S_ = 1 + N * T
            _, dz = hip.layernorm_bwd(dy.view(-1, D), self.tok, m.norm.weight, VIT_EPS, dtok, g, b_, accumulate=False,
                                      emit=dict(mode=hip.EMIT_ROWS, rows=B * S_, dtype=self.saved[-1]["dt"], scale=self.saved[-1]["drop_m"], group=S_))
        else:
            hip.layernorm_bwd(dy.view(-1, D), self.tok, m.norm.weight, VIT_EPS, dtok, g, b_, accumulate=False)
        del dy
        bank = m.blocks[0]._bank()
        dt_run = self.saved[-1]["dt"]
        banked = bank is not None and dt_run != torch.float32 and all(sv["merged"] for sv in self.saved)
        ws = bank.workspace(dt_run, dout.device) if banked else torch.zeros((len(m.blocks), D * D + D), dtype=torch.float32, device=dout.device)
        # Data parallel: in the pretraining / retrieval models this node is the LAST one autograd runs (created first, its input needs
        # no gradient), so every gradient outside the visual encoder is final now and can be exchanged while the ViT backward (half of
        # the backward pass) runs; the blocks' own gradients follow four blocks at a time (alpro_amd.dist.grads_final -> FlatAdamW).
        # That is only true when no other anchored backward is still pending (tr.Anchor sets others_pending; a model that anchors the
        # text encoder first runs this node BEFORE it): then nothing is declared final here and everything goes at synchronize().
        from alpro_amd import dist
        overlap = dist.collectives_active() and not getattr(self, "others_pending", False)
        if overlap:
            dist.grads_final(all_but=list(self.enc.parameters()))
        nb = len(m.blocks)
        group_hi = nb
        for i, (blk, sv) in enumerate(zip(reversed(m.blocks), reversed(self.saved))):
            sv["ws"] = ws[nb - 1 - i]
            sv["defer_product_rule"] = banked
            dtok, dz = blk.backward(sv, dtok, dz=dz, emit_for=self.saved[nb - 2 - i] if i + 1 < nb else None)
            if (i + 1) % 4 == 0 or i + 1 == nb:   # a group of blocks is through: their merged-projection product rule in batched launches ...
                lo = nb - 1 - i
                if banked:
                    bank.product_rule(lo, group_hi, dt_run)
                if overlap and i + 1 < nb:       # ... and then their gradients are final (the last group goes at synchronize())
                    dist.grads_final(params=[p for b in m.blocks[lo:group_hi] for p in b.parameters()] + (list(m.norm.parameters()) if group_hi == nb else []))
                group_hi = lo
            sv.clear()
        m._embed_backward(self.rows, dtok, B, T, N, self.Wg)
        self.saved = self.rows = self.tok = None
        return None  # pixels need no gradient
