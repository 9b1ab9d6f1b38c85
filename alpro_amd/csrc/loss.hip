// Row-wise softmax cross-entropy with the gradient written straight in the GEMM operand dtype: the MLM loss of
// alpro_models.py:368-371 over (B*Lt, 30522) fp32 logits (312 MB at B=64).  One workgroup per row, two sweeps over
// the row (max / sum-exp, then (softmax - onehot) * scale); ignore_index rows contribute 0 and get a zero gradient.
#include "common.hpp"

namespace alpro {
namespace {

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void xent_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels, int ignore_index,
                                                   float* __restrict__ loss_rows, T* __restrict__ dlogits, int64_t ldd,
                                                   const float* __restrict__ grad_scale, int V, int Vpad) {
  __shared__ float sh[4];
  const int m = blockIdx.x;
  const float* x = logits + (int64_t)m * ld;
  const int64_t label = labels[m];
  const bool valid = label != ignore_index;
  float mx = -INFINITY;
  for (int i = threadIdx.x * 2; i < V; i += 512) {
    if (i + 1 < V) {
      const float2 v = *(const float2*)(x + i);  // rows are 8-byte aligned (ld even)
      mx = fmaxf(mx, fmaxf(v.x, v.y));
    } else {
      mx = fmaxf(mx, x[i]);
    }
  }
  mx = block_reduce(mx, sh, true);
  float s = 0.f;
  for (int i = threadIdx.x * 2; i < V; i += 512) {
    if (i + 1 < V) {
      const float2 v = *(const float2*)(x + i);
      s += expf(v.x - mx) + expf(v.y - mx);
    } else {
      s += expf(x[i] - mx);
    }
  }
  s = block_reduce(s, sh, false);
  const float lse = mx + logf(s);
  if (threadIdx.x == 0) loss_rows[m] = valid ? lse - x[label] : 0.f;
  if (dlogits) {
    const float gs = valid ? *grad_scale : 0.f;
    T* d = dlogits + (int64_t)m * ldd;
    for (int i = threadIdx.x * 2; i < Vpad; i += 512) {
      float g0 = 0.f, g1 = 0.f;
      if (i < V) g0 = (expf(x[i] - lse) - (i == label ? 1.f : 0.f)) * gs;
      if (i + 1 < V) g1 = (expf(x[i + 1] - lse) - (i + 1 == label ? 1.f : 0.f)) * gs;
      if constexpr (sizeof(T) == 4) {
        d[i] = g0;
        if (i + 1 < Vpad) d[i + 1] = g1;
      } else {
        *(uint32_t*)(d + i) = pack2(g0, g1, (T*)0);  // Vpad and ldd are even
      }
    }
  }
}
}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_softmax_xent(const float* logits, int64_t ld, const int64_t* labels, int ignore_index, float* loss_rows, void* dlogits,
                                  int dtype, int64_t ldd, const float* grad_scale, int M, int V, int Vpad, void* stream) {
  ALPRO_CHECK(logits && labels && loss_rows && M > 0 && V > 0, "alpro_softmax_xent: bad args");
  ALPRO_CHECK(ld % 2 == 0 && ((uintptr_t)logits % 8) == 0, "alpro_softmax_xent: logits rows must be 8-byte aligned");
  ALPRO_CHECK(!dlogits || (grad_scale && Vpad >= V && Vpad % 2 == 0 && ldd >= Vpad && ldd % 2 == 0), "alpro_softmax_xent: bad gradient buffer");
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(xent_kernel<T>, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, ignore_index, loss_rows, (T*)dlogits, ldd, grad_scale, V, Vpad));
  return check_launch("alpro_softmax_xent");
}
