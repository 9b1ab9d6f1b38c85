/*
 * alpro_hip.h -- C ABI of libalpro_hip.so: the MI355X (gfx950) kernels behind ALPRO's
 * video-text hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer owned
 * by the caller (torch allocates), `stream` is a hipStream_t passed as void*, every entry point
 * is asynchronous on `stream` (no device- or stream-wide synchronisation anywhere) and re-entrant.
 * Workspaces are caller-provided.  State the library keeps: the option table, and per (device,
 * stream) that launches the persistent GEMM a pair of tile-scheduler counter blocks -- from the
 * caller (alpro_hip_set_sched_workspace) or, by default, from ONE lazily made hipMalloc of 135 KB
 * per device (the only allocation this library makes; see alpro_hip_set_sched_workspace below).
 * Return value: 0 = ok, nonzero = error (text via alpro_hip_last_error()).
 *
 * The reference (salesforce/ALPRO) is pure Python: the "FFI" these entry points replace is the
 * set of torch/ATen calls its nn.Modules make on this path.  Each declaration cites the reference
 * call sites it stands in for (paths relative to the reference root).  INTEGRATION.md shows
 * the ctypes binding.
 *
 * Storage dtypes: GEMM/attention operands are ALPRO_BF16 (throughput), ALPRO_F16, or ALPRO_F32
 * (exact mode: fp32 MFMA, used to prove parity at 1e-3 on VTC logits).  The residual stream,
 * LayerNorm statistics, softmax, accumulators, losses and all gradients w.r.t. parameters are
 * always fp32.
 */
#ifndef ALPRO_HIP_H
#define ALPRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALPRO_HIP_ABI_VERSION 20

enum { ALPRO_OK = 0, ALPRO_ERR_INVALID = 1, ALPRO_ERR_LAUNCH = 2 };
enum { ALPRO_F32 = 0, ALPRO_BF16 = 1, ALPRO_F16 = 2 };
/* GELU_BWD (backward of a GELU Linear, vit.py:61 / xbert.py:423): C = (alpha*acc + bias) * gelu'(C2[m, n]) where C2 is
 * the READ-ONLY pre-activation the forward GEMM saved; fuses the elementwise backward into the dgrad GEMM. */
enum { ALPRO_ACT_NONE = 0, ALPRO_ACT_GELU = 1, ALPRO_ACT_RELU = 2, ALPRO_ACT_GELU_BWD = 3,
       /* round 3: the GELU Linear's forward writes gelu'(pre-activation) into C2 instead of the pre-activation (GELU_SAVE_GRAD, C2 required),
        * and its dgrad multiplies by the saved value (MUL_SAVED: C = (alpha*acc + bias) * C2[m, n], C2 read only) -- the ~13 VALU operations
        * per element of recomputing gelu' in the backward epilogue (the slowest dgrad of the model in round 2) become one multiply. */
       ALPRO_ACT_GELU_SAVE_GRAD = 4, ALPRO_ACT_MUL_SAVED = 5 };

/* Row maps: how GEMM/LayerNorm row m addresses the (B, 1 + N*T, D) token tensor whose patch token
 * (n, t) lives at row 1 + n*T + t of its clip (vit.py:147 'b (h w t) m').
 *   IDENTITY        row = m
 *   SKIP_CLS        p0 = N*T.  m enumerates x[:, 1:]           -> row = m + m / p0 + 1     (vit.py:146,162)
 *   FRAME_TOKENS    p0 = T, p1 = N.  m = (b*T + t)*(N+1) + j   -> j == 0: CLS of clip b (gather) /
 *                   side-buffer row b*T+t (scatter); j > 0: row = b*(1+N*T) + 1 + (j-1)*T + t
 *                                                                              (vit.py:165-172,184-196)
 *   PATCH_EMBED     p0 = T, p1 = N.  m = (b*T + t)*N + n       -> out row = b*(1+N*T) + 1 + n*T + t,
 *                   residual row = n*T + t (a (N*T, D) table)              (vit.py:233-239,342,349-361)
 */
enum { ALPRO_MAP_IDENTITY = 0, ALPRO_MAP_SKIP_CLS = 1, ALPRO_MAP_FRAME_TOKENS = 2, ALPRO_MAP_PATCH_EMBED = 3 };

const char* alpro_hip_last_error(void);
int alpro_hip_abi_version(void);
/* Measurement knobs (never change results): "gemm_tile" (128 / 256 forces a tile kernel, 0 = heuristic), "gemm_grid" (cap of the
 * persistent grid), "gemm_tune" (DMA issue placement variant / ablations), "tn_splits" (number of token ranges of the weight-gradient GEMM), "tn_kind" (1: weight-gradient workgroups return before their
 * epilogue -- timing only, results are garbage).  Defaults come from the
 * environment (ALPRO_GEMM_TILE, ...) once at load time; there is no reference counterpart (tools/ and bench.py use it). */
int alpro_hip_set_option(const char* name, int value);
/* Round 5: the same knobs scoped to ONE stream (value >= 0 sets the override for launches on `stream`, value < 0 removes it; launches on
 * other streams keep the process-wide value).  The only option the library consults per stream today is "cu_budget" -- the number of
 * compute units a launch may count on while a collective's kernels hold the rest (alpro_amd.optim sets it on the stream its backward runs
 * on while the overlapped gradient exchange is in flight, run_pretrain_sparse.py:432,601); "gemm_sched" (1 = the persistent NT GEMM takes
 * its tiles from per-XCD ticket counters, 0 = the static round-robin walk) is a process-wide A/B knob.  No reference counterpart. */
int alpro_hip_set_stream_option(void* stream, const char* name, int value);
/* Round 6: who owns the tile scheduler's memory (no reference counterpart: ATen's allocator owns everything there).  The persistent 8-phase GEMM
 * keeps two counter blocks per (device, stream) -- launch n draws its tile tickets from block n & 1 and clears the other.
 *   alpro_hip_sched_workspace_bytes()            size of one such pair (2112 bytes).
 *   alpro_hip_set_sched_workspace(stream, p, n)  use the caller's device memory p (n >= that size, 16-byte aligned, alive until the stream is
 *                                                released) for launches on `stream` of the current device; cleared by a memset ON the stream.
 *                                                p == NULL is alpro_hip_release_stream.  A process that registers a workspace for every stream
 *                                                it launches GEMMs on makes the library allocation-free.
 *   alpro_hip_release_stream(stream)             drop the stream's slot (64 per process; a further stream runs the static tile walk: same
 *                                                results, no tolerance to CU theft) and its option overrides.  For a stream that is idle and
 *                                                about to be destroyed: the runtime may give a later stream the same handle value.
 * Without a registered workspace the first persistent launch on a device makes one hipMalloc of 64 pairs (135 KB) and clears the launching
 * stream's pair with hipMemsetAsync on that stream -- no hipDeviceSynchronize (rounds 1-5 had one there). */
size_t alpro_hip_sched_workspace_bytes(void);
int alpro_hip_set_sched_workspace(void* stream, void* ptr, size_t bytes);
int alpro_hip_release_stream(void* stream);

/* Round 6: the temporal half's qkv Linear AND its attention in one launch (forward only; vit.py:84-98 Attention.forward on the
 * 'b (h w t) m -> (b h w) t m' view of vit.py:152-156): out[m, h*64 + d] = sum_j softmax_j((q_m . k_j) * scale) v_j[d] over the T rows j of row m's
 * frame group (rows g*T .. g*T + T - 1), with q | k | v = A W^T + bias computed per (256-row tile, head) and consumed out of the accumulators:
 * the (M, 3*H*64) tensor is never written.  A (M, K) and W (3*H*64, K; rows [q | k | v], head-major like Attention.qkv.weight) K-contiguous
 * 16-bit, bias (3*H*64) fp32 or NULL, out (M, H*64) 16-bit.  M % 32 == 0, T in {1, 2, 4, 8, 16}, K % 128 == 0.  Same roundings as
 * alpro_gemm + alpro_attn_temporal_fwd (q, k, v, P and the output to 16 bits; everything else fp32); the two paths agree to the order of
 * their fp32 sums.  Training: qkv_out (M, 3*H*64; row stride ldq) != NULL also receives q | k | v exactly as alpro_gemm would have stored them,
 * and lse != NULL the per-row log-sum-exp in alpro_attn_temporal_fwd's layout ((32-row chunk * H + h) * 32 + row in chunk) -- what
 * alpro_attn_temporal_bwd reads; the saving is then the attention launch and its re-read of q | k | v, not the write. */
int alpro_gemm_qkv_tattn(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int dtype,
                         int M, int H, int T, int K, float scale, void* qkv_out, int64_t ldq, float* lse, void* stream);

/* ---------------------------------------------------------------------------------------------
 * C[map(m), n] = residual[map(m), n] + row_scale[m / row_scale_group] * act(alpha * sum_k A[m,k] W[n,k] + bias[n])
 * A (M,K) and W (N,K) are K-contiguous in `dtype`; accumulation is fp32 on MFMA
 * (v_mfma_f32_32x32x16_{bf16,f16} / v_mfma_f32_32x32x2_f32).  K % (128 / sizeof(dtype)) == 0.
 * Replaces every nn.Linear on the path: vit.py:60-63 (Mlp), :84,:98 (Attention.qkv/proj), :161
 * (temporal_fc), :230-239 (PatchEmbed conv as GEMM over alpro_patchify rows); xbert.py:273-294
 * (query/key/value, fused as one N=3*H GEMM), :357, :422, :435, :659, :681; alpro_models.py:38-39,42,66-71.
 * The residual add / drop_path scale / GELU that follow those Linears in the reference
 * (vit.py:157-162,181-196,212; vit_utils.py:137-151) are fused here.
 */
typedef struct {
  const void* A;
  const void* W;
  void* C;
  int64_t lda, ldw, ldc;
  int M, N, K;
  int dtype;   /* storage dtype of A and W */
  int c_dtype; /* dtype of C: `dtype` or ALPRO_F32 */
  float alpha;
  const float* bias;      /* (N) fp32 or NULL */
  int act;                /* ALPRO_ACT_* */
  const float* row_scale; /* fp32 or NULL; entry m / row_scale_group */
  int row_scale_group;
  const float* residual;  /* fp32 or NULL */
  int64_t ldr;
  int map_mode, map_p0, map_p1; /* ALPRO_MAP_* applied to C rows and residual rows */
  float* side;            /* FRAME_TOKENS: (B*T, N) fp32 buffer receiving the j == 0 rows (no residual) */
  int64_t ld_side;
  void* C2;               /* GELU/RELU: optional (M, N) `dtype` copy of the pre-activation alpha*acc+bias (kept for the
                             backward).  GELU_BWD: the saved pre-activation, read only (required).  GELU_SAVE_GRAD: receives
                             gelu'(pre-activation) (required).  MUL_SAVED: the saved factor, read only (required). */
  int64_t ldc2;
  float drop_p;           /* dropout on the value BEFORE the residual add (xbert.py:358,436): keep iff hash(seed, m*N+n) */
  uint32_t drop_seed;     /* passes, kept values scaled by 1/(1-p); 0 = off.  Identity map only. */
  const float* bias2;     /* optional (N) fp32 added AFTER the row scale: C = residual + row_scale*act(..) + bias2.  SKIP_CLS map
                             only: the merged temporal projection W_fc*W_proj of vit.py:157-162, where drop_path scales the
                             proj output but not temporal_fc's own bias. */
  int64_t m_off;          /* round 4: absolute index of row 0 for what the epilogue indexes by row -- row_scale[(m_off + m) / group] and the
                             dropout hash (m_off + m) * N + n.  0 for callers; the library sets it on the second launch when it splits a
                             ragged M into whole 256-row tiles + a remainder. */
  int32_t c2_tiled;       /* round 5: C2 is in the TILE layout of the 8-phase kernel instead of (M, ldc2) rows -- only for the pair
                             ALPRO_ACT_GELU_SAVE_GRAD (writes gelu') / ALPRO_ACT_MUL_SAVED (reads it back) of one (M, N) output shape, and
                             only when alpro_gemm_c2_tiled_rows(M, N, K, dtype) > 0: C2 then holds that many rows of N 16-bit elements
                             (M rounded up to whole 256-row tiles) whose order is private to the library.  ldc2 is ignored. */
  int32_t reserved0;
} alpro_gemm_desc_t;

/* Rows of the (rows, N) 16-bit buffer a tile-layout C2 needs for an (M, N, K) GEMM of `dtype` operands with contiguous operands (lda = ldw = K,
 * ldc = N), or 0 when this shape would not run on the 8-phase kernel's packed epilogue under the current options (then keep c2_tiled = 0 and the
 * row layout).  The same answer must hold for the forward (GELU_SAVE_GRAD) and the backward (MUL_SAVED) launch: both have the same (M, N). */
int64_t alpro_gemm_c2_tiled_rows(int64_t M, int64_t N, int64_t K, int dtype);

int alpro_gemm(const alpro_gemm_desc_t* d, void* stream);

/* njobs independent GEMMs in ONE launch (round 3): the same descriptor, once in host memory (validated there) and once as a device-resident
 * array the kernel reads (blockIdx.y = job).  All jobs share the operand dtype and use plain epilogues (alpha, bias, row scale, fp32 residual,
 * identity map); 128 x 128 tiles.  Used for the per-block 768^3 products of the merged temporal projection (W_e = W_fc W_p after every
 * optimizer step; dW_fc += dW_e W_p^T and dW_p += W_fc^T dW_e in backward; vit.py:157-162 under autograd): 12 ViT blocks x 3 products that
 * each fill a seventh of the chip when launched one by one. */
int alpro_gemm_batch(const alpro_gemm_desc_t* descs_host, const alpro_gemm_desc_t* descs_device, int njobs, void* stream);

/* The small per-block terms of the merged temporal projection, all blocks in one launch (jobs in DEVICE memory, D == 768):
 *   mode 0: b1[n] = sum_m wfc[n, m] bp[m]                                  (W_fc b_p, the merged bias; after an optimizer step)
 *   mode 1: g_fc[n, m] += db1[n] bp[m];  g_bp[m] += sum_n wfc[n, m] db1[n]   (product rule of the bias path; single writer, fixed order) */
typedef struct alpro_tproj_job {
  const float* wfc; /* (D, D) temporal_fc.weight */
  const float* bp;  /* (D) temporal_attn.proj.bias */
  float* b1;        /* mode 0 output (D) */
  const float* db1; /* mode 1: gradient w.r.t. b1 (D) */
  float* g_fc;      /* mode 1: temporal_fc.weight.grad (D, D), accumulated */
  float* g_bp;      /* mode 1: temporal_attn.proj.bias.grad (D), accumulated */
} alpro_tproj_job_t; /* 48 bytes */
int alpro_tproj_small(const alpro_tproj_job_t* jobs_device, int njobs, int D, int mode, void* stream);

/* y[m] = LayerNorm(x[map(m)]) * gamma + beta over D == 768 fp32 inputs; writes `y` in y_dtype and,
 * if non-NULL, an fp32 copy y32 plus mean/rstd (rows) for the backward pass.
 * Replaces nn.LayerNorm at vit.py:154,180,200,372 (eps 1e-6) and xbert.py:211,359,437,661 (eps 1e-12);
 * the rearrange/cat copies of vit.py:147,165-172 become the gather map. */
int alpro_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* y,
                        int y_dtype, int64_t ldy, float* y32, float* mean, float* rstd, int rows, int D,
                        int map_mode, int map_p0, int map_p1, void* stream);

/* Residual add fused into the LayerNorm that follows it (vit.py:162 -> :180, :196 -> :200, :212 -> next block's :154; xbert.py:358-359,
 * 436-437).  The producing Linear writes only its output `delta` (operand dtype, plain (rows, 768) row order, drop-path / dropout already
 * applied by the GEMM epilogue); this kernel computes x' = x_in + gathered delta (+ delta_bias on the rows that receive a delta), stores
 * x' to x_out (fp32; may be NULL when nobody needs it, may alias x_in) and LayerNorm(x') to y (`dtype`) [and y32, fp32, 16-bit modes
 * only].  Replaces the fp32 read-modify-write of x in the GEMM epilogue, the CLS side buffer and alpro_cls_mean_residual.
 *   ALPRO_ADD_IDENTITY      rows = M:              v = x_in[m] + delta[m]                                  -> x_out[m], y[m]
 *   ALPRO_ADD_PRE_SPATIAL   rows = B*T*(N+1), p0 = T, p1 = N, m = (b*T+t)*(N+1)+j, r = token row of (b, j, t):
 *                           j > 0: v = x_in[r] + delta[r - b - 1] (+ bias)   (delta in x[:, 1:] order);  j = 0: v = x_in[r]
 *                                                                                                          -> x_out[r], y[m]
 *   ALPRO_ADD_PRE_MLP       rows = B*(1+N*T), r = b*S + k: k > 0: v = x_in[r] + delta[(b*T+t)*(N+1)+1+n];
 *                           k = 0: v = x_in[r] + mean_t delta[(b*T+t)*(N+1)]                               -> x_out[r], y[r]
 *   ALPRO_ADD_PRE_TEMPORAL  rows = B*(1+N*T): v = x_in[r] + delta[r] -> x_out[r]; k > 0: y[r - b - 1] = LN(v) */
enum { ALPRO_ADD_IDENTITY = 0, ALPRO_ADD_PRE_SPATIAL = 1, ALPRO_ADD_PRE_MLP = 2, ALPRO_ADD_PRE_TEMPORAL = 3 };
int alpro_add_layernorm_fwd(const float* x_in, const void* delta, int dtype, const float* delta_bias, int add_mode, float* x_out,
                            const float* gamma, const float* beta, float eps, void* y, float* y32, int64_t rows, int D, int p0, int p1,
                            void* stream);
/* ALPRO_ADD_PRE_MLP with the temporal branch's add deferred into it (round 6; the no-grad forward): v = x_in[r] + delta_t[r - b - 1] +
 * delta_t_bias + delta_s[(b*T+t)*(N+1)+1+n] for patch rows (in this order: bit for bit what ALPRO_ADD_PRE_SPATIAL followed by ALPRO_ADD_PRE_MLP
 * store), the CLS row as in ALPRO_ADD_PRE_MLP.  The caller runs ALPRO_ADD_PRE_SPATIAL with x_out = NULL before it, so the intermediate
 * x + temporal branch (vit.py:162) is never written.  rows = B*(1+N*T); x_out may alias x_in. */
int alpro_add_layernorm_pre_mlp2(const float* x_in, const void* delta_t, const float* delta_t_bias, const void* delta_s, int dtype, float* x_out,
                                 const float* gamma, const float* beta, float eps, void* y, int64_t rows, int D, int T, int N, void* stream);

/* Divided space-time attention, temporal half (vit.py:146-157 -> Attention.forward :81-96):
 * rows = B*N*T tokens in (b, n, t) order, each group of T consecutive rows attends within itself.
 * qkv (rows, 3*H*64) as written by the qkv Linear, out (rows, H*64) == 'transpose(1,2).reshape'.
 * T must divide 32.  softmax(q k^T * scale) v on MFMA with a block-diagonal group mask. */
int alpro_attn_temporal_fwd(const void* qkv, void* out, int dtype, int64_t rows, int T, int H, float scale,
                            float* lse /* optional (ceil(rows/32), H, 32) row log-sum-exp for the backward */, void* stream);

/* Precise CLS-query attention of the 16-bit operand modes (round 4; DESIGN.md section 2, "CLS rows precise"): for every (sequence, head) of
 * the spatial half (vit.py:180 on (B*T, 1+N) tokens, query row 0 = the CLS token of :165-167) or of a text-mode BERT layer (xbert.py:299-341,
 * query row 0 = [CLS]):   out[s, h*64:(h+1)*64] = softmax(q_cls K^T * scale + key_bias) V   in fp32, where q_cls and the CLS token's own
 * k / v come UNROUNDED from qkv_cls (batch / group rows of 3*H*64 fp32: sequences s*group .. s*group+group-1 share row s -- the T frame
 * copies of one clip; group = 1 for text) and the other tokens' K / V from the 16-bit qkv tensor (batch*L, 3*H*64) that alpro_gemm wrote.
 * drop_p / drop_seed: attention-probability dropout with the mask alpro_attn_fwd draws for query 0 (xbert.py:331).  out (batch, H*64) fp32. */
int alpro_attn_cls_fwd(const void* qkv, int dtype, const float* qkv_cls, const float* key_bias /* (batch, L) or NULL */, float* out, int batch,
                       int L, int H, int group, float scale, float drop_p, uint32_t drop_seed, void* stream);

/* The Linears (and pre-LayerNorms) of the precise CLS-row side path: C[M, N] = residual + row_scale[m] * act(LN(A)[M, K] W[N, K]^T + bias), all
 * fp32 (exact fp32 MFMA, fixed summation order), for M = B or B*T rows against a full (N, K) fp32 master weight (vit.py:84,98,59-65 / xbert.py
 * :304-316,357,421,435 applied to the CLS rows only).  ln_gamma / ln_beta (K,) or NULL: LayerNorm of the A rows over their K elements fused
 * into the operand load (vit.py:180 norm1, :200 norm2).  act: ALPRO_ACT_NONE / ALPRO_ACT_GELU (exact erf).  N % 16 == 0, K % 64 == 0. */
int alpro_gemm_rows_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int M, int N, int K,
                        const float* bias /* (N) or NULL */, int act, const float* row_scale /* (M) or NULL */,
                        const float* residual /* (M, ldr) or NULL */, int64_t ldr, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        void* stream);

/* Full (bidirectional) attention over `batch` sequences of L <= 256 tokens, head_dim 64:
 * spatial half of divided attention (vit.py:180 on (B*T, 1+N) tokens, :81-96) and the BERT
 * text / fusion self-attention (xbert.py:299-341) where key_bias (batch, L) is the additive
 * (1 - mask) * -10000 of xbert.py:936-937 (NULL = no mask).  K/V of one (sequence, head) stay
 * resident in LDS; QK^T and PV run on MFMA; softmax in fp32 registers.  lse (batch, H, L) optional. */
int alpro_attn_fwd(const void* qkv, void* out, int dtype, int batch, int L, int H, float scale,
                   const float* key_bias, float* lse, float drop_p, uint32_t drop_seed,
                   /* round 4, precise CLS query fused into the same launch (NULL = off; semantics of alpro_attn_cls_fwd with K / V taken from the
                    * images already staged in LDS): cls_q (batch / cls_group, 3*H*64) fp32, cls_out (batch, H*64) fp32; 16-bit dtypes only */
                   const float* cls_q, int cls_group, float* cls_out, void* stream);
/* drop_p > 0: dropout on the attention probabilities (xbert.py:331), mask = hash(seed, ((b*H+h)*L+q)*L+key). */

/* out[(b*T+t)*N + n, c*256 + i*16 + j] = img[b, t, c, ph*16 + i, pw*16 + j], n = ph*(W/16) + pw:
 * the im2col rows of the stride-16 Conv2d (vit.py:230-238), cast to `dtype`. */
int alpro_patchify(const float* img, void* out, int dtype, int BT, int C, int Himg, int Wimg, void* stream);

/* Clip preparation in one pass over the raw pixels: ImageNorm (src/datasets/data_utils.py:437-457, applied to the three clip tensors at
 * dataloader.py:104-115) fused with the MPM random-erase crop (dataset_pretrain_sparse.py:277-311).  raw (B, T, 3, H, W) uint8 or fp32;
 * boxes (B, 4) int32 {top, left, h, w} on the DEVICE (NULL: no crop / context); mean3 / std3 are HOST arrays; scale = 1/255 when the
 * pixels are 0..255 and the mean is <= 1 (the reference's test), else 1.
 *   visual = (x*scale - mean)/std;  crop = the same of (x inside the box, 0 outside);  context = of (0 inside, x outside).
 * crop / context may be NULL. */
int alpro_prepare_clips(const void* raw, int raw_is_u8, const int* boxes, float scale, const float* mean3, const float* std3, float* visual,
                        float* crop, float* context, int B, int T, int H, int W, void* stream);

/* Round 5: the input of the fusion encoder as a gather of sequences (alpro_models.py:278-281, 325-330, 360-363: torch.cat of text and video
 * embeddings, with `text_embeds[neg_text]` / `video_embeds[neg_video]` for the hard negatives).  Sequence s of the (S, Lt + Lv, D) fusion batch
 * is text-pool sequence ti[s] followed by video-pool sequence vi[s]:
 *   fwd: out32[s] = [text[ti[s]] ; video[vi[s]]] (fp32) and, if out_t != NULL, the same rows in `dtype` (the first fusion layer's GEMM operand).
 *   bwd: dtext[p] = sum_{s: ti[s] == p} (d32[s, :Lt] + d_t[s, :Lt]),  dvideo[p] = sum_{s: vi[s] == p} (d32[s, Lt:] + d_t[s, Lt:]), s ascending
 *        (d_t: optional 16-bit part of the gradient; every pool row is written, rows nobody used with zeros).  D == 768. */
int alpro_gather_seq_fwd(const float* text, const float* video, const int64_t* ti, const int64_t* vi, float* out32, void* out_t, int dtype, int S,
                         int Lt, int Lv, int D, void* stream);
int alpro_gather_seq_bwd(const float* d32, const void* d_t, int dtype, const int64_t* ti, const int64_t* vi, float* dtext, float* dvideo, int S, int Pt,
                         int Pv, int Lt, int Lv, int D, void* stream);

/* x_out[b, 0, :] = x_in[b, 0, :] + mean_t side[b*T + t, :]   (vit.py:184-187,195-196) */
int alpro_cls_mean_residual(const float* x_in, int64_t ld_batch_in, const float* side, float* x_out,
                            int64_t ld_batch_out, int B, int T, int D, void* stream);

/* Final LayerNorm + temporal mean pool (vit.py:372 + :484-492): x (B, 1+N*T, D) ->
 * out32 (B, 1+N, D) fp32 [and out_t in `dtype` if non-NULL]. */
int alpro_vit_final_pool(const float* x, const float* gamma, const float* beta, float eps, float* out32,
                         void* out_t, int dtype, int B, int T, int N, int D, void* stream);

/* BERT embeddings (xbert.py:186-213): word[ids] + type[0] + pos[l] -> LayerNorm -> y32 (+ y_t). */
int alpro_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                         const float* gamma, const float* beta, float eps, float* y32, void* y_t, int dtype,
                         float* mean, float* rstd, int rows, int L, int D, float drop_p, uint32_t drop_seed,
                         void* stream);
/* drop_p > 0: the embedding dropout of xbert.py:212 on the LayerNorm output, mask = hash(seed, m*D+n). */

/* dst[i] = (dtype) src[i]: parameter / activation cast used when the storage dtype is 16-bit. */
int alpro_cast_from_f32(const float* src, void* dst, int dtype, int64_t n, void* stream);

/* ---- backward of the same path (what autograd derives for the reference: loss.backward() at
 * run_pretrain_sparse.py:599 through vit.py:136-213 and xbert.py:457-519) --------------------------------- */

/* dQKV (rows, 3*H*64) from dO (rows, H*64), the saved qkv / out and the row log-sum-exp of the forward.
 * P is recomputed on MFMA; delta = rowsum(dO o O); see attention_bwd.hip. */
int alpro_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int dtype,
                   int batch, int L, int H, float scale, const float* key_bias, float drop_p, uint32_t drop_seed,
                   void* stream);
int alpro_attn_temporal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int dtype,
                            int64_t rows, int T, int H, float scale, void* stream);

/* LayerNorm backward over D == 768: dx[map(m)] (+)= dLN(dy[m] (+ dy2[m]), x[map(m)]); dgamma/dbeta are ACCUMULATED
 * into fp32 buffers.  The forward gather map becomes a scatter; a row gathered more than once (the clip's CLS row under
 * FRAME_TOKENS, once per frame) receives T terms -- that map requires accumulate = 1.
 * Column sums (dgamma, dbeta, the emit's colsum_pre) and those T terms -- REDUCTION WORKSPACE (ABI 16), the same contract in
 * alpro_gather_cast / alpro_sumsq:
 *   workspace != NULL (16-byte aligned; >= 9216 bytes, plus rows / (N + 1) * 3072 bytes under FRAME_TOKENS; 2048 * 9216 bytes + that
 *   never cap the grid): every workgroup writes its partial sums to its own slot, every frame's CLS term is parked in its own row, and
 *   two small kernels add slots / terms in a fixed order -> results are bit-reproducible run to run;
 *   workspace == NULL: fp32 atomics (one per column per workgroup; the CLS terms straight into the row): no extra memory, the last bits
 *   vary with the arrival order.
 *   The buffer is scratch: used in stream order, contents meaningless afterwards, may be shared by every call on one stream. */
int alpro_layernorm_bwd(const void* dy, int dy_dtype, int64_t ld_dy, const float* dy2, const float* x, int64_t ldx,
                        const float* gamma, float eps, float* dx, int64_t ld_dx, int accumulate, float* dgamma,
                        float* dbeta, int rows, int D, int map_mode, int map_p0, int map_p1, float drop_p,
                        uint32_t drop_seed, void* workspace, size_t workspace_bytes, void* stream);
/* drop_p > 0: the incoming gradient dy (+dy2) is first multiplied by the dropout mask hash(seed, m*D+n)/(1-p) that the
 * forward applied to this LayerNorm's OUTPUT (embedding dropout). */

/* alpro_layernorm_bwd that ALSO emits the finished gradient row as the `dy_dtype` operand row(s) of the GEMMs that consume it next --
 * what a following alpro_gather_cast would produce from re-reading dx (autograd has no counterpart: the reference materialises the
 * rearranged / scaled gradients as separate tensors, vit.py:147,160,165-172,184-196 under autograd).  v = finished dx row of token row r:
 *   ALPRO_EMIT_ROWS      out[r] = dropout(v; emit_drop_p, emit_drop_seed, index r*D+c) * emit_scale[r / group]
 *                        emit_extra_cls = B (with map_mode SKIP_CLS, p0 = T, p1 = N): the B CLS rows, which that LayerNorm never touches,
 *                        are emitted too (read from dx)
 *   ALPRO_EMIT_FRAME     r = b*(1+N*T) + k: k > 0 -> out[(b*T+t)*(N+1)+1+n] = v * emit_scale[b*T+t]; k = 0 -> the T rows (b*T+t)*(N+1),
 *                        each v * emit_scale[b*T+t] / T
 *   ALPRO_EMIT_SKIP_CLS  k > 0 -> out[r-b-1] = v * emit_scale[(r-b-1) / group]; emit_colsum_pre[c] += v[c] (unscaled); shared (CLS) rows
 *                        emit nothing
 * Rows accumulated by several source rows (the CLS row under the FRAME_TOKENS map) are never emitted.  ld_dx must be D. */
enum { ALPRO_EMIT_NONE = 0, ALPRO_EMIT_ROWS = 1, ALPRO_EMIT_FRAME = 2, ALPRO_EMIT_SKIP_CLS = 3 };
int alpro_layernorm_bwd_emit(const void* dy, int dy_dtype, int64_t ld_dy, const float* dy2, const float* x, int64_t ldx,
                             const float* gamma, float eps, float* dx, int64_t ld_dx, int accumulate, float* dgamma, float* dbeta,
                             int rows, int D, int map_mode, int map_p0, int map_p1, float drop_p, uint32_t drop_seed, void* emit_out,
                             int emit_dtype /* dy_dtype, or any dtype when dy is fp32 */, int emit_mode, int emit_p0, int emit_p1, const float* emit_scale, int emit_scale_group, float emit_drop_p,
                             uint32_t emit_drop_seed, float* emit_colsum_pre, int emit_extra_cls, void* workspace, size_t workspace_bytes,
                             void* stream);

/* out[c, r] = in[r, c] (r < R), 0 for R <= r < Rpad: puts the token dimension last so that dgrad / wgrad run on
 * the NT GEMM (dX = dY (W^T)^T, dW = dY^T (X^T)^T).  `in` is fp32 or out_dtype.  colsum (C) fp32, optional:
 * column sums are atomically accumulated into it (the bias gradient). */
int alpro_transpose(const void* in, int in_dtype, int64_t ld_in, void* out, int out_dtype, int64_t ld_out, int R, int C,
                    int Rpad, float* colsum, void* stream);

/* njobs independent fp32 -> out_dtype transposes (same semantics as alpro_transpose without colsum) in one launch: the W^T operands of
 * all Linear layers after an optimizer step.  `jobs` is DEVICE memory; tile0 = number of 64 x 64 tiles of the jobs before this one
 * (a tile grid of ceil(C / 64) x ceil(Rpad / 64) per job), total_tiles = their sum over all jobs. */
typedef struct alpro_transpose_job {
  const float* in;
  void* out;
  int64_t ld_in, ld_out;
  int32_t R, C, Rpad, tile0;
} alpro_transpose_job_t; /* 48 bytes */
int alpro_transpose_batch(const alpro_transpose_job_t* jobs, int njobs, int total_tiles, int out_dtype, void* stream);

/* out[m, :] = (dtype)(row_scale[m / group] * src[map(m), :]) over D == 768: turns the fp32 token-gradient tensor into
 * the operand rows of the backward GEMMs (inverse of the forward scatter maps, drop-path scale re-applied).  Under
 * FRAME_TOKENS the j == 0 rows read the clip's CLS row times cls_scale (= 1/T, the frame mean of vit.py:187). */
int alpro_gather_cast(const float* src, int64_t ld, void* out, int dtype, int rows, int D, int map_mode, int map_p0,
                      int map_p1, const float* row_scale, int row_scale_group, float cls_scale, float drop_p,
                      uint32_t drop_seed, float* colsum, float* colsum_pre, void* workspace, size_t workspace_bytes, void* stream);
/* workspace: reduction workspace for colsum / colsum_pre (see alpro_layernorm_bwd; >= 6144 bytes, 512 * 6144 never caps the grid), NULL = atomics.
 * drop_p > 0: additionally re-applies the GEMM-epilogue dropout mask hash(seed, m*D+n)/(1-p) (backward of alpro_gemm's drop_p).
 * colsum (optional, (D) fp32): colsum[n] += sum_m out[m, n] -- the bias gradient of the Linear these rows are the dY of;
 * colsum_pre (optional): the same column sums taken BEFORE the row scale (bias gradient of a Linear that sits after the
 * drop-path scale, e.g. temporal_fc under the merged temporal projection). */

/* du = dh * gelu'(u) with the erf GELU (vit.py:61 / xbert.py:423 backward). */
int alpro_gelu_bwd(const void* dh, const void* u, void* du, int dtype, int64_t n, void* stream);

/* dside[b*T+t, :] = dx_out[b, 0, :] / T  (backward of alpro_cls_mean_residual w.r.t. the parked CLS rows). */
int alpro_cls_mean_bwd(const float* dx_out, int64_t ld_batch, float* dside, int B, int T, int D, void* stream);

/* dst[idx[i], :] += src[i, :] (idx NULL: row i % idx_mod): embedding-table gradients (xbert.py:203-210 backward).  Rows whose index equals
 * skip_idx (>= 0) are dropped: nn.Embedding(padding_idx = pad_token_id) keeps the pad row's gradient at zero (xbert.py:171); -1 = none.
 * idx == NULL and skip_idx < 0 (the position table): one wave per destination row adds its rows in ascending order -- bit-reproducible.
 * With idx: fp32 atomics, the duplicates' order varies run to run (the reproducible caller sorts: alpro_amd/hip.py scatter_add_rows). */
int alpro_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int rows, int idx_mod, int D, int64_t skip_idx, void* stream);
/* Round 5: the indexed form in a FIXED order without torch's sort-based index_put_: dst[idx[i], :] += src[i, :], duplicates added in ascending
 * i, one writer per destination row (dst_rows < 2^19; rows <= 8192: the keys are sorted by one workgroup in LDS).  keys_ws: 8192 uint32 of
 * scratch.  Rows with idx == skip_idx are dropped (-1 = none), and so are rows whose idx lies outside [0, dst_rows): nothing is ever
 * written outside the table (torch's index_put_, which this replaces, device-asserts on such an index). */
int alpro_scatter_add_rows_ordered(const float* src, const int64_t* idx, float* dst, int rows, int D, int64_t dst_rows, int64_t skip_idx, uint32_t* keys_ws,
                                   void* stream);

/* Weight gradient without transposed copies: C[N, K] (fp32, ATOMICALLY accumulated) += A[M, N]^T B[M, K], A = dY and
 * B = X row-major in a 16-bit dtype, contraction over tokens split across the grid (gemm_tn.hip).  C must be
 * initialised (zero or the running gradient).  fp32 operands: use alpro_transpose + alpro_gemm.
 * colsum (optional, (N) fp32, atomically accumulated): colsum[n] += sum_m A[m, n] -- the bias gradient of the same Linear,
 * taken from the dY fragments the kernel already holds (nn.Linear backward: grad_bias = grad_output.sum(0)). */
int alpro_gemm_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int dtype, int M,
                      int N, int K, float* colsum, void* stream);

/* The same product with the partial results of the token ranges combined through a caller-provided workspace instead of fp32
 * atomics: each range stores its 256 x 256 partial tiles with plain 16-byte stores and a second launch adds them to C (and the
 * bias-gradient partials to colsum) in a fixed order -- bit-reproducible run to run, and the partials cross the fabric once as
 * plain stores instead of as read-modify-write atomics (0.50 -> 0.47 ms on the 100416-token MLP weight gradients).
 * workspace: device memory, 16-byte aligned, >= alpro_gemm_tn_workspace_bytes(M, N, K) bytes (0 when the library would not split
 * the tokens: then workspace may be NULL and the call is alpro_gemm_tn_acc); it is scratch -- contents are dead when the call's
 * work has run, so one buffer can serve every call of a stream.  workspace == NULL: exactly alpro_gemm_tn_acc. */
size_t alpro_gemm_tn_workspace_bytes(int M, int N, int K);
/* Round 5: the token-range plan depends on the compute units the launch may count on (option "cu_budget" of the launch stream), which the
 * size query above cannot know: it answers with the largest workspace any budget's plan needs.  alpro_gemm_tn_ranges gives the number of
 * token ranges of the workspace plan for a given CU count (256 = the whole chip) -- for tests and the tuning tools. */
int alpro_gemm_tn_ranges(int M, int N, int K, int compute_units);
int alpro_gemm_tn_acc_ws(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int dtype, int M,
                         int N, int K, float* colsum, void* workspace, size_t workspace_bytes, void* stream);

/* out[n] (fp32) += sum_m A[m, n]: bias gradients. */
int alpro_colsum_acc(const void* A, int64_t lda, float* out, int dtype, int M, int N, void* stream);

/* Masked-LM cross entropy (alpro_models.py:368-371: CrossEntropyLoss over (B*Lt, vocab), ignore_index -100):
 * loss_rows[m] = logsumexp(logits[m]) - logits[m, label] (0 for ignored rows) and, if dlogits != NULL,
 * dlogits[m, :Vpad] = (softmax - onehot) * *grad_scale in `dtype` (zero for ignored rows and for columns >= V), i.e.
 * already the operand of the decoder's backward GEMMs.  *grad_scale is a device scalar (1 / #valid rows). */
int alpro_softmax_xent(const float* logits, int64_t ld, const int64_t* labels, int ignore_index, float* loss_rows,
                       void* dlogits, int dtype, int64_t ldd, const float* grad_scale, int M, int V, int Vpad,
                       void* stream);

/* Video-text contrastive loss over gathered features (alpro_models.py:103-128 Pretrain, :570-587 Prompter, :750-779 Retrieval; the
 * normalised (B, E) projections v / t are this rank's rows, gv / gt the (G = world * B, E) all-gathered ones, positives at columns
 * col0 + i (col0 = local_rank * B, :121-123), temp the learnable temperature (clamped to [0.001, 0.5] like :80-81):
 *   sim_v2t = v gt^T / temp, sim_t2v = t gv^T / temp   (B, G) fp32, written out (hard-negative mining :287-306 reads them)
 *   *loss   = (mean_i CE(sim_v2t[i], col0 + i) + mean_i CE(sim_t2v[i], col0 + i)) / 2;  lse (2B): row log-sum-exps for the backward.
 * _bwd: gradients w.r.t. v, t (B, E), gv, gt (G, E) and temp (*dtemp, optional) for upstream *dloss; ds_* are (B, G) scratch.
 * *loss and *dtemp are summed in a fixed order by one finishing workgroup each (ABI 16; before: one fp32 atomic per row) -- bit-reproducible. */
int alpro_vtc_loss_fwd(const float* v, const float* t, const float* gv, const float* gt, const float* temp, int B, int G, int E, int col0,
                       float* sim_v2t, float* sim_t2v, float* lse, float* loss, void* stream);
int alpro_vtc_loss_bwd(const float* v, const float* t, const float* gv, const float* gt, const float* temp, int B, int G, int E, int col0,
                       const float* sim_v2t, const float* sim_t2v, const float* lse, const float* dloss, float* ds_v2t, float* ds_t2v,
                       float* dv, float* dt, float* dgv, float* dgt, float* dtemp, void* stream);

/* ---- step epilogue on flat fp32 buffers (run_pretrain_sparse.py:633-648, src/optimization/adamw.py:40-103) ---- */

/* *out += sum(x[i]^2): global gradient norm for clip_grad_norm_.  workspace: reduction workspace (see alpro_layernorm_bwd; one float per
 * workgroup, 8192 bytes never cap the grid) -> fixed summation order; NULL = one fp32 atomic per workgroup. */
int alpro_sumsq(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* HF-style AdamW over n contiguous parameters: g' = g * grad_scale * min(1, max_norm / (sqrt(*gnorm_sq)*grad_scale + 1e-6));
 * m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; p -= step_size * m / (sqrt(v) + eps); p -= lr * wd * p.
 * step_size = lr * sqrt(1-b2^t) / (1-b1^t) is computed by the caller (correct_bias).  gnorm_sq NULL or max_norm <= 0: no clip.
 * dyn_state (optional, DEVICE, 4 floats {loss scale S, growth tracker, applied steps, skipped steps}): dynamic loss scaling for fp16
 * gradient operands -- what apex.amp does for the reference when its configs set fp16 = 1 (run_pretrain_sparse.py:441,596-634;
 * apex/amp/scaler.py is not vendored: restated from its documented behaviour).  With dyn_state the kernel (i) skips the whole update
 * when *gnorm_sq is not finite (an fp16 overflow somewhere in the backward), (ii) divides the gradients by S when grads_scaled != 0,
 * (iii) takes the bias-correction step count from dyn_state[2] (applied steps only; `step_size` is then ignored, correct_bias selects
 * the formula).  gnorm_sq is required with dyn_state. */
int alpro_adamw_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float step_size, const float* gnorm_sq, float max_norm,
                     float grad_scale, const float* dyn_state, int grads_scaled, int correct_bias,
                     int zero_grad /* round 4: also clear g (optimizer.zero_grad(), run_pretrain_sparse.py:648, folded in) */, void* stream);
/* ... and the same pass ALSO refreshing the 16-bit mirror of the parameters (round 6; lp: n values of lp_dtype = ALPRO_BF16 / ALPRO_F16, the flat
 * copy the GEMM operands are views of; rounded exactly as alpro_cast_from_f32 rounds; NULL = alpro_adamw_step).  A skipped step (overflow)
 * leaves the mirror alone, like the parameters.  Replaces the separate cast launch behind the step: 6 B / parameter less traffic. */
int alpro_adamw_step_lp(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                        float step_size, const float* gnorm_sq, float max_norm, float grad_scale, const float* dyn_state, int grads_scaled,
                        int correct_bias, int zero_grad, void* lp, int lp_dtype, void* stream);

/* After alpro_adamw_step on the same stream: *gnorm_sq not finite -> S = max(S * backoff, min_scale), tracker = 0, skipped += 1;
 * else applied += 1, tracker += 1 and after `window` clean steps S = min(S * growth, max_scale).  apex defaults: growth 2, backoff 0.5,
 * window 2000, initial S 2^16, max 2^24. */
int alpro_loss_scale_update(float* dyn_state, const float* gnorm_sq, float growth, float backoff, int window, float min_scale,
                            float max_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALPRO_HIP_H */
