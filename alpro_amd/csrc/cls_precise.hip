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
