// Step epilogue of the data-parallel training loop on flat fp32 buffers: global gradient norm and the
// reference's HF-style AdamW (src/optimization/adamw.py:40-103) with the clip_grad_norm_ coefficient
// (run_pretrain_sparse.py:633) folded in.  One pass over (param, grad, m, v): 16 B/param read + 12 B/param written.
#include "common.hpp"
#include <algorithm>

namespace alpro {
namespace {

// part == nullptr: out += sum(x^2) through one fp32 atomic per workgroup (order varies run to run).  part given: workgroup b leaves its sum in
// part[b] and sumsq_finish_kernel adds the slots in a fixed order -- the squared norm feeds the clip coefficient of EVERY parameter, so a
// last-bit difference here would make two identical steps differ everywhere.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out, float* __restrict__ part) {
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 v = *(const float4*)(x + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int64_t k = i; k < n; ++k) s += x[k] * x[k];
    }
  }
  s = wave_sum(s);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = sh[0] + sh[1] + sh[2] + sh[3];
    if (part) part[blockIdx.x] = t;
    else atomicAdd(out, t);
  }
}

// out += sum of part[0 .. n): one workgroup, thread t adds slots t, t + 256, ... in ascending order, then a fixed tree.
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum(s);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out += (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// LP: storage type of the optional 16-bit mirror of the parameters (the GEMM operands' flat copy, alpro_adamw_step_lp; float = none).
template <typename LP>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                                    float weight_decay, float step_size, const float* __restrict__ gnorm_sq,
                                                    float max_norm, float grad_scale, const float* __restrict__ dyn, int grads_scaled,
                                                    int correct_bias, int zero_grad, LP* __restrict__ lp) {
  float coef = grad_scale;
  if (dyn) {
    // dynamic loss scaling (fp16 operands; apex.amp semantics, run_pretrain_sparse.py:596-634 with fp16 = 1): dyn = {loss scale S,
    // growth tracker, completed optimizer steps}.  The gradients hold S * dL/dw (unless the caller already unscaled them); a non-finite
    // squared norm means some fp16 gradient operand overflowed: the whole update is skipped -- parameters, moments and the 16-bit mirror
    // untouched -- and alpro_loss_scale_update halves S.  The bias correction uses the DEVICE step counter, which only counts applied updates.
    if (!isfinite(*gnorm_sq)) {   // skipped step: nothing moves -- but a caller that asked for consumed gradients still gets them zeroed
      if (zero_grad) {
        const int64_t stride0 = (int64_t)gridDim.x * blockDim.x * 4;
        for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride0)
          for (int k = 0; k < 4 && i + k < n; ++k) g[i + k] = 0.f;
      }
      return;
    }
    if (grads_scaled) coef /= dyn[0];
    const float t = dyn[2] + 1.0f;
    step_size = correct_bias ? lr * sqrtf(1.0f - powf(beta2, t)) / (1.0f - powf(beta1, t)) : lr;
  }
  if (gnorm_sq && max_norm > 0.f) {
    const float total = sqrtf(*gnorm_sq) * coef;
    coef *= fminf(max_norm / (total + 1e-6f), 1.0f);  // torch.nn.utils.clip_grad_norm_
  }
  auto update = [&](float (&pv)[4], const float (&gv)[4], float (&mv)[4], float (&vv)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gv[k] * coef;
      mv[k] = mv[k] * beta1 + (1.0f - beta1) * gr;
      vv[k] = vv[k] * beta2 + (1.0f - beta2) * gr * gr;
      const float denom = sqrtf(vv[k]) + eps;
      pv[k] = pv[k] - step_size * (mv[k] / denom);
      if (weight_decay > 0.f) pv[k] = pv[k] - lr * weight_decay * pv[k];
    }
  };
  auto store4 = [&](int64_t i, const float (&pv)[4], const float (&mv)[4], const float (&vv)[4]) {
    *(float4*)(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    *(float4*)(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (zero_grad) *(float4*)(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);   // optimizer.zero_grad() folded in (round 4): no separate 0.94 GB memset
    if constexpr (sizeof(LP) == 2) {
      if (lp) *(u32x2*)(lp + i) = mk2(pack2(pv[0], pv[1], (LP*)0), pack2(pv[2], pv[3], (LP*)0));   // the same roundings as alpro_cast_from_f32
    }
  };
  // Two float4 of each array per iteration (round 6): all eight loads are issued before the first divide -- one float4 per array left the pass at
  // 5.1 TB/s with three quarters of the wave cycles stalled on memory.
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += 2 * stride) {
    const int64_t j = i + stride;
    if (j + 4 <= n) {   // (i + 4 <= n as well)
      const float4 a0 = *(const float4*)(p + i), b0 = *(const float4*)(g + i), c0 = *(const float4*)(m + i), d0 = *(const float4*)(v + i);
      const float4 a1 = *(const float4*)(p + j), b1 = *(const float4*)(g + j), c1 = *(const float4*)(m + j), d1 = *(const float4*)(v + j);
      float pv0[4] = {a0.x, a0.y, a0.z, a0.w}, mv0[4] = {c0.x, c0.y, c0.z, c0.w}, vv0[4] = {d0.x, d0.y, d0.z, d0.w};
      float pv1[4] = {a1.x, a1.y, a1.z, a1.w}, mv1[4] = {c1.x, c1.y, c1.z, c1.w}, vv1[4] = {d1.x, d1.y, d1.z, d1.w};
      const float gv0[4] = {b0.x, b0.y, b0.z, b0.w}, gv1[4] = {b1.x, b1.y, b1.z, b1.w};
      update(pv0, gv0, mv0, vv0);
      store4(i, pv0, mv0, vv0);
      update(pv1, gv1, mv1, vv1);
      store4(j, pv1, mv1, vv1);
      continue;
    }
    for (int64_t q = i; q < n && q <= j; q += stride) {   // the last one or two chunks of this thread, element by element where they are ragged
      const int cnt = (q + 4 <= n) ? 4 : (int)(n - q);
      float pv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < cnt; ++k) { pv[k] = p[q + k]; gv[k] = g[q + k]; mv[k] = m[q + k]; vv[k] = v[q + k]; }
      update(pv, gv, mv, vv);
      if (cnt == 4) {
        store4(q, pv, mv, vv);
      } else {
        for (int k = 0; k < cnt; ++k) {
          p[q + k] = pv[k]; m[q + k] = mv[k]; v[q + k] = vv[k];
          if (zero_grad) g[q + k] = 0.f;
          if constexpr (sizeof(LP) == 2) { if (lp) lp[q + k] = from_f32<LP>(pv[k]); }
        }
      }
    }
  }
}

// One thread: the loss-scale schedule of apex's dynamic LossScaler (scale_window 2000, halve on overflow, double after a window of
// clean steps, clamped), entirely on the device -- no host sync in the training loop.
__global__ void loss_scale_update_kernel(float* __restrict__ dyn, const float* __restrict__ gnorm_sq, float growth, float backoff,
                                         float window, float min_scale, float max_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!isfinite(*gnorm_sq)) {
    dyn[0] = fmaxf(dyn[0] * backoff, min_scale);
    dyn[1] = 0.f;
    dyn[3] += 1.f;  // skipped steps
  } else {
    dyn[2] += 1.f;  // applied steps
    dyn[1] += 1.f;
    if (dyn[1] >= window) {
      dyn[0] = fminf(dyn[0] * growth, max_scale);
      dyn[1] = 0.f;
    }
  }
}

inline int grid_for(int64_t n) {
  int64_t g = (n / 4 + 255) / 256;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}
}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_sumsq(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ALPRO_CHECK(x && out && n > 0, "alpro_sumsq: bad args");
  ALPRO_CHECK(((uintptr_t)x % 16) == 0, "alpro_sumsq: x must be 16-byte aligned");
  ALPRO_CHECK(!workspace || (((uintptr_t)workspace % 4) == 0 && workspace_bytes >= sizeof(float)), "alpro_sumsq: bad workspace");
  int grid = grid_for(n);
  float* part = (float*)workspace;
  if (part) grid = (int)std::min<size_t>((size_t)grid, workspace_bytes / sizeof(float));
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, out, part);
  if (part) hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, grid, out);
  return check_launch("alpro_sumsq");
}

extern "C" int alpro_adamw_step_lp(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, float step_size, const float* gnorm_sq, float max_norm, float grad_scale,
                                   const float* dyn_state, int grads_scaled, int correct_bias, int zero_grad, void* lp, int lp_dtype, void* stream) {
  ALPRO_CHECK(p && g && m && v && n > 0, "alpro_adamw_step: bad args");
  ALPRO_CHECK(!dyn_state || gnorm_sq, "alpro_adamw_step: dynamic loss scaling needs the squared gradient norm (overflow detection)");
  ALPRO_CHECK(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
              "alpro_adamw_step: buffers must be 16-byte aligned");
  ALPRO_CHECK(!lp || ((lp_dtype == ALPRO_BF16 || lp_dtype == ALPRO_F16) && ((uintptr_t)lp % 8) == 0), "alpro_adamw_step_lp: the mirror is a 16-bit, 8-byte-aligned buffer");
  const dim3 grid(grid_for(n)), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define ALPRO_ADAMW_GO(LP_, ptr)                                                                                                          \
  hipLaunchKernelGGL(adamw_kernel<LP_>, grid, blk, 0, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_size, gnorm_sq, max_norm, \
                     grad_scale, dyn_state, grads_scaled, correct_bias, zero_grad, ptr)
  if (lp && lp_dtype == ALPRO_BF16) ALPRO_ADAMW_GO(bf16_t, (bf16_t*)lp);
  else if (lp) ALPRO_ADAMW_GO(f16_t, (f16_t*)lp);
  else ALPRO_ADAMW_GO(float, (float*)nullptr);
#undef ALPRO_ADAMW_GO
  return check_launch("alpro_adamw_step");
}

extern "C" int alpro_adamw_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, float step_size, const float* gnorm_sq, float max_norm, float grad_scale,
                                const float* dyn_state, int grads_scaled, int correct_bias, int zero_grad, void* stream) {
  return alpro_adamw_step_lp(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_size, gnorm_sq, max_norm, grad_scale, dyn_state, grads_scaled,
                             correct_bias, zero_grad, nullptr, ALPRO_F32, stream);
}

extern "C" int alpro_loss_scale_update(float* dyn_state, const float* gnorm_sq, float growth, float backoff, int window, float min_scale,
                                       float max_scale, void* stream) {
  ALPRO_CHECK(dyn_state && gnorm_sq && growth >= 1.f && backoff > 0.f && backoff <= 1.f && window > 0 && min_scale > 0.f && max_scale >= min_scale,
              "alpro_loss_scale_update: bad args");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dyn_state, gnorm_sq, growth, backoff, (float)window, min_scale, max_scale);
  return check_launch("alpro_loss_scale_update");
}
