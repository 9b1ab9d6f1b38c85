// Precise CLS-row side path of the 16-bit operand modes (round 4).
//
// Why: BASELINE.json's north star asks for VTC logits within 1e-3 of the fp32 reference.  The VTC features are projections of ONE row per
// sequence -- the CLS token of the last ViT block / of the last text-mode BERT layer.  A rounding-site model of the fp16 path
// (tools/precision_model.py, profiles/r4_precision_model.txt) shows where its 7e-4 rms logit error comes from: the patch / word rows'
// rounding errors reach a CLS row only through attention, averaged over 197 x T keys; the CLS row's OWN roundings -- LayerNorm output ->
// q/k/v -> attention output -> projection -> MLP, 24 branch additions deep -- are not averaged by anything.  Evaluating exactly those rows
// in fp32 (B x T rows per ViT block, B per BERT layer: < 1e-3 of the FLOPs) removes 2/3 of the error (7.0e-4 -> 2.0e-4 rms in the model).
//
// alpro_attn_cls_fwd is the one piece of that chain that is not a small fp32 GEMM / LayerNorm (alpro_gemm / alpro_layernorm_fwd on (B, 768)
// fp32 rows): softmax(q_cls K^T * scale + bias) V for the CLS query of every (sequence, head) with
//   * q, and the CLS token's own k / v, taken UNROUNDED from `qkv_cls` (fp32, one row per CLS token: the T frame copies of a clip share it),
//   * the other tokens' K / V read from the 16-bit qkv tensor the big GEMM produced (their errors are the averaged kind),
//   * scores, softmax (online, log2 domain) and the P V accumulation in fp32 on the VALU, attention-probability dropout with the same
//     (seed, b, h, q = 0, key) hash as alpro_attn_fwd.
// One wave per (sequence, head): lane = (key slot g = lane >> 3, 8-element head chunk e = lane & 7), so a wave instruction reads 8 whole
// 128-byte K (or V) rows; the 8 key slots keep independent online-softmax states that are merged once at the end.  HBM-bound on re-reading
// K and V of every head once (0.31 GB per ViT block at B = 64 x 8 frames).
#include "common.hpp"

namespace alpro {
namespace {

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;

template <typename T>
__global__ __launch_bounds__(256) void attn_cls_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_cls, const float* __restrict__ key_bias,
                                                       float* __restrict__ out, int batch, int L, int H, int group, float scale, float drop_p,
                                                       uint32_t drop_seed) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= batch * H) return;
  const int s = unit / H, h = unit - s * H;
  const int g = lane >> 3, e = lane & 7;
  const int64_t ldq = 3 * (int64_t)H * HD;
  const float* cls = qkv_cls + (int64_t)(s / group) * ldq + h * HD + e * 8;
  float q[8], kc[8], vc[8];
  {
    const float sl = scale * LOG2E;
    const float4 a = *(const float4*)cls, b = *(const float4*)(cls + 4);
    q[0] = a.x * sl; q[1] = a.y * sl; q[2] = a.z * sl; q[3] = a.w * sl; q[4] = b.x * sl; q[5] = b.y * sl; q[6] = b.z * sl; q[7] = b.w * sl;
    const float4 c = *(const float4*)(cls + H * HD), d = *(const float4*)(cls + H * HD + 4);
    kc[0] = c.x; kc[1] = c.y; kc[2] = c.z; kc[3] = c.w; kc[4] = d.x; kc[5] = d.y; kc[6] = d.z; kc[7] = d.w;
    const float4 f = *(const float4*)(cls + 2 * H * HD), w = *(const float4*)(cls + 2 * H * HD + 4);
    vc[0] = f.x; vc[1] = f.y; vc[2] = f.z; vc[3] = f.w; vc[4] = w.x; vc[5] = w.y; vc[6] = w.z; vc[7] = w.w;
  }
  const T* base = qkv + (int64_t)s * L * ldq + h * HD + e * 8;
  const float* kb = key_bias ? key_bias + (int64_t)s * L : nullptr;
  const uint32_t th = drop_thresh24(drop_p);
  const float ks = drop_seed ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t drop_base = (((uint64_t)s * H + h) * L) * (uint64_t)L;   // query 0 of (s, h): the index alpro_attn_fwd hashes
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const int nit = (L + 7) >> 3;
#pragma unroll 4
  for (int it = 0; it < nit; ++it) {
    const int j = it * 8 + g;
    const bool live = j < L;
    float kf[8], vf[8];
    if (live && j > 0) {
      const u32x4 kr = *(const u32x4*)(base + (int64_t)j * ldq + H * HD);
      const u32x4 vr = *(const u32x4*)(base + (int64_t)j * ldq + 2 * H * HD);
      unpack_chunk<T>(kr, kf);
      unpack_chunk<T>(vr, vf);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { kf[i] = kc[i]; vf[i] = vc[i]; }
    }
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = fmaf(q[i], kf[i], d);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (kb && live) d = fmaf(kb[j], LOG2E, d);
    if (live) {   // (uniform over the 8 lanes of a key slot)
      const float mn = fmaxf(m, d);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);   // exp2(-inf) == 0 on the first key of the slot
      const float p = __builtin_amdgcn_exp2f(d - mn);
      l = fmaf(l, alpha, p);
      float pd = p;
      if (drop_seed) pd = drop_keep(drop_seed, drop_base + (uint64_t)j, th) ? p * ks : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(acc[i], alpha, pd * vf[i]);
      m = mn;
    }
  }
  // merge the 8 key slots (lanes with equal e): global maximum, rescale, sum
  float M = m;
  M = fmaxf(M, __shfl_xor(M, 8, 64));
  M = fmaxf(M, __shfl_xor(M, 16, 64));
  M = fmaxf(M, __shfl_xor(M, 32, 64));
  const float w = __builtin_amdgcn_exp2f(m - M);   // a slot that saw no key (L < 8): m = -inf -> weight 0
  l *= w;
  l += __shfl_xor(l, 8, 64);
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float a = acc[i] * w;
    a += __shfl_xor(a, 8, 64);
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    acc[i] = a * inv;
  }
  if (g == 0) {
    float* o = out + (int64_t)s * H * HD + h * HD + e * 8;
    *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *(float4*)(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}


// ------------------------------------------------------------------------------------------------
// alpro_gemm_rows_f32: the Linears of the precise side path -- C[M, N] = residual + row_scale * act(LN?(A)[M, K] W[N, K]^T + bias), fp32
// throughout (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation), for a HANDFUL of rows (M = B or B*T) against a full weight
// matrix.  alpro_gemm's 128x128 fp32 tile turns such a problem into N/128 workgroups walking the whole K serially (fc2 of the CLS rows:
// 6 workgroups x 96 K-steps = 190 us on an otherwise idle chip); here a workgroup owns 64 rows x 16 columns, its four waves split K four
// ways and meet in LDS (fixed summation order: bit-reproducible), so the same problem is 48 workgroups x 12 steps (~10 us).
// Optional fused LayerNorm of the A rows (K == the row width: statistics of the 64 rows recomputed per workgroup, two-pass in registers).
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int ACT, bool LN, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_rows_f32_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
                                                                float* __restrict__ C, int64_t ldc, int M, int N, int K, const float* __restrict__ bias,
                                                                const float* __restrict__ row_scale, const float* __restrict__ residual, int64_t ldr,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float (*part)[64][16] = (float (*)[64][16])smem_f;          // [NW][64][16] partial sums of the K slices
  float (*stat)[2] = (float (*)[2])(smem_f + NW * 64 * 16);   // [64][2] mean, rstd of the tile's rows (fused LayerNorm)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
  const int r15 = lane & 15, kg = lane >> 4;
  if (LN && NW == 8 && K == 768) {
    // round 6: the hidden size on 8 waves -- eight lanes per row, the lane's 24 float4 loaded ONCE (all in flight together: one memory latency
    // instead of 2 x 48 loads per lane in dependent batches, which was ~half of the launch's 25 us) and both passes taken from registers
    const int r = tid >> 3, q = tid & 7;
    const float* a = A + (int64_t)min(m0 + r, M - 1) * lda + q * 4;
    float4 t[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) t[i] = *(const float4*)(a + i * 32);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) sum += (t[i].x + t[i].y) + (t[i].z + t[i].w);
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    const float mean = sum / 768.0f;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const float d0 = t[i].x - mean, d1 = t[i].y - mean, d2 = t[i].z - mean, d3 = t[i].w - mean;
      sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    if (q == 0) { stat[r][0] = mean; stat[r][1] = rsqrtf(sq / 768.0f + eps); }
    __syncthreads();
  } else if (LN) {   // statistics of the 64 rows: four lanes per row (a quarter of the row each, float4 loads), two passes (mean, then centred squares)
    if (wave < 4) {
      const int r = wave * 16 + (lane >> 2), q = lane & 3;
      const float* a = A + (int64_t)min(m0 + r, M - 1) * lda + q * (K >> 2);
      const int nv = K >> 4;   // float4 per lane
      float sum = 0.f;
      for (int i = 0; i < nv; ++i) {
        const float4 t = *(const float4*)(a + i * 4);
        sum += (t.x + t.y) + (t.z + t.w);
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      const float mean = sum / (float)K;
      float sq = 0.f;
      for (int i = 0; i < nv; ++i) {
        const float4 t = *(const float4*)(a + i * 4);
        const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
        sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      if (q == 0) { stat[r][0] = mean; stat[r][1] = rsqrtf(sq / (float)K + eps); }
    }
    __syncthreads();
  }
  float mu[4], rs[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) { mu[f] = LN ? stat[f * 16 + r15][0] : 0.f; rs[f] = LN ? stat[f * 16 + r15][1] : 1.f; }
  const int kslice = K / NW, k0 = wave * kslice;
  const float* ap[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) ap[f] = A + (int64_t)min(m0 + f * 16 + r15, M - 1) * lda + k0 + kg * 4;
  const float* wp = W + (int64_t)(n0 + r15) * ldw + k0 + kg * 4;
  const float* gp = LN ? gamma + k0 + kg * 4 : nullptr;
  const float* bp = LN ? beta + k0 + kg * 4 : nullptr;
  f32x4v acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc[f] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int steps = kslice >> 4;
#pragma unroll 3
  for (int s = 0; s < steps; ++s) {
    const float4 b4 = *(const float4*)(wp + s * 16);
    float4 a4[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) a4[f] = *(const float4*)(ap[f] + s * 16);
    if (LN) {
      const float4 g4 = *(const float4*)(gp + s * 16), o4 = *(const float4*)(bp + s * 16);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        a4[f].x = fmaf((a4[f].x - mu[f]) * rs[f], g4.x, o4.x);
        a4[f].y = fmaf((a4[f].y - mu[f]) * rs[f], g4.y, o4.y);
        a4[f].z = fmaf((a4[f].z - mu[f]) * rs[f], g4.z, o4.z);
        a4[f].w = fmaf((a4[f].w - mu[f]) * rs[f], g4.w, o4.w);
      }
    }
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float aa[4] = {a4[f].x, a4[f].y, a4[f].z, a4[f].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[e], bb[e], acc[f], 0, 0, 0);
    }
  }
  // D layout: lane holds rows kg*4 + r, column r15 of each 16x16 fragment
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][f * 16 + kg * 4 + r][r15] = acc[f][r];
  __syncthreads();
  if (tid < 256) {
    const int row = tid >> 2, c4 = (tid & 3) * 4;
    const int m = m0 + row, n = n0 + c4;
    if (m < M) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NW; ++w)   // fixed order: bit-reproducible
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += part[w][row][c4 + e];
      const float sc = row_scale ? row_scale[m] : 1.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e] + (bias ? bias[n + e] : 0.f);
        if (ACT == ALPRO_ACT_GELU) x = gelu_erf(x);
        x *= sc;
        if (residual) x += residual[(int64_t)m * ldr + n + e];
        v[e] = x;
      }
      *(float4*)(C + (int64_t)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}


struct RowsArgs {
  const float* A; int64_t lda; const float* W; int64_t ldw; float* C; int64_t ldc; int M, N, K;
  const float* bias; const float* row_scale; const float* residual; int64_t ldr; const float* gamma; const float* beta; float eps;
};
template <int ACT, bool LN, int NW>
void launch_rows_inst(const RowsArgs& a, hipStream_t st) {
  constexpr int LDS = (NW * 64 * 16 + 128) * (int)sizeof(float);
  static DeviceOnce once;
  once.run([&] { (void)hipFuncSetAttribute((const void*)gemm_rows_f32_kernel<ACT, LN, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
  hipLaunchKernelGGL((gemm_rows_f32_kernel<ACT, LN, NW>), dim3(a.N / 16, (a.M + 63) / 64), dim3(NW * 64), LDS, st, a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M,
                     a.N, a.K, a.bias, a.row_scale, a.residual, a.ldr, a.gamma, a.beta, a.eps);
}
template <int NW>
void launch_rows_nw(const RowsArgs& a, bool ln, bool gelu, hipStream_t st) {
  if (ln) {
    if (gelu) launch_rows_inst<ALPRO_ACT_GELU, true, NW>(a, st); else launch_rows_inst<ALPRO_ACT_NONE, true, NW>(a, st);
  } else {
    if (gelu) launch_rows_inst<ALPRO_ACT_GELU, false, NW>(a, st); else launch_rows_inst<ALPRO_ACT_NONE, false, NW>(a, st);
  }
}

}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_attn_cls_fwd(const void* qkv, int dtype, const float* qkv_cls, const float* key_bias, float* out, int batch, int L, int H,
                                  int group, float scale, float drop_p, uint32_t drop_seed, void* stream) {
  ALPRO_CHECK(qkv && qkv_cls && out && batch > 0 && H > 0 && L > 0 && group > 0, "alpro_attn_cls_fwd: bad args");
  ALPRO_CHECK(batch % group == 0, "alpro_attn_cls_fwd: batch=%d is not a whole number of groups of %d sequences", batch, group);
  ALPRO_CHECK(dtype == ALPRO_BF16 || dtype == ALPRO_F16, "alpro_attn_cls_fwd: the 16-bit qkv tensor of a 16-bit operand mode (the fp32 mode has no rounding to repair)");
  ALPRO_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)qkv_cls % 16) == 0 && ((uintptr_t)out % 16) == 0, "alpro_attn_cls_fwd: pointers must be 16-byte aligned");
  ALPRO_CHECK(!drop_seed || (drop_p > 0.f && drop_p < 1.f), "alpro_attn_cls_fwd: dropout needs 0 < p < 1");
  const int units = batch * H;
  const dim3 grid((units + 3) / 4), block(256);
  if (dtype == ALPRO_BF16)
    hipLaunchKernelGGL(attn_cls_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)qkv, qkv_cls, key_bias, out, batch, L, H, group, scale, drop_p, drop_seed);
  else
    hipLaunchKernelGGL(attn_cls_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (const f16_t*)qkv, qkv_cls, key_bias, out, batch, L, H, group, scale, drop_p, drop_seed);
  return check_launch("alpro_attn_cls_fwd");
}

extern "C" int alpro_gemm_rows_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int M, int N, int K,
                                   const float* bias, int act, const float* row_scale, const float* residual, int64_t ldr,
                                   const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream) {
  ALPRO_CHECK(A && W && C && M > 0 && N > 0 && K > 0, "alpro_gemm_rows_f32: bad args");
  ALPRO_CHECK(N % 16 == 0 && K % 64 == 0, "alpro_gemm_rows_f32: N=%d must be a multiple of 16 and K=%d of 64", N, K);
  ALPRO_CHECK(lda % 4 == 0 && ldw % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0,
              "alpro_gemm_rows_f32: rows must be 16-byte aligned");
  ALPRO_CHECK(act == ALPRO_ACT_NONE || act == ALPRO_ACT_GELU, "alpro_gemm_rows_f32: act %d unsupported (none / gelu)", act);
  ALPRO_CHECK(!ln_gamma == !ln_beta, "alpro_gemm_rows_f32: LayerNorm needs both gamma and beta");
  ALPRO_CHECK(!ln_gamma || K % 64 == 0, "alpro_gemm_rows_f32: fused LayerNorm needs K a multiple of 64 (got %d)", K);
  // K split over the waves of a workgroup: 16 waves for the long contractions (K = 3072: 12 steps per wave), 8 below (K = 768: 6 steps)
  const int nw = (K >= 2048 && K % 256 == 0) ? 16 : (K % 128 == 0 ? 8 : 4);
  const RowsArgs ra{A, lda, W, ldw, C, ldc, M, N, K, bias, row_scale, residual, ldr, ln_gamma, ln_beta, ln_eps};
  hipStream_t st = (hipStream_t)stream;
  const bool ln = ln_gamma != nullptr, ge = act == ALPRO_ACT_GELU;
  if (nw == 16) launch_rows_nw<16>(ra, ln, ge, st);
  else if (nw == 8) launch_rows_nw<8>(ra, ln, ge, st);
  else launch_rows_nw<4>(ra, ln, ge, st);
  return check_launch("alpro_gemm_rows_f32");
}
