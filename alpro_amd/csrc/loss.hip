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
  if (!valid) {
    // an ignored row (85 % of the MLM rows) contributes no loss and a zero gradient row: written without reading the 122 KB of logits three times
    // (the sweeps below give the same +0.0 values for finite logits: (exp(..) - 0) * 0)
    if (threadIdx.x == 0) loss_rows[m] = 0.f;
    if (dlogits) {
      T* d = dlogits + (int64_t)m * ldd;
      for (int i = threadIdx.x * 2; i < Vpad; i += 512) {
        if constexpr (sizeof(T) == 4) {
          d[i] = 0.f;
          if (i + 1 < Vpad) d[i + 1] = 0.f;
        } else {
          *(uint32_t*)(d + i) = 0u;
        }
      }
    }
    return;
  }
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

// ---- video-text contrastive loss (alpro_models.py:103-128 / 570-587 / 750-779) ----------------------------------------------------
// sim_v2t = v gt^T / temp, sim_t2v = t gv^T / temp over the GATHERED features (G = world * B rows), soft-target cross entropy with
// the positives at columns col0 + i, both directions averaged: loss = (mean_i ce_v[i] + mean_i ce_t[i]) / 2.  B x G x 256 is tiny
// (33 MFLOP at B = 64, G = 512): the point of the kernel is ONE launch instead of ~25 ATen launches, not throughput.
// Block (dir, i): query row in LDS, thread j owns key columns j, j + 256, ...
constexpr int VTC_MAX_E = 1024;

__global__ __launch_bounds__(256) void vtc_fwd_kernel(const float* __restrict__ v, const float* __restrict__ t, const float* __restrict__ gv,
                                                      const float* __restrict__ gt, const float* __restrict__ temp, int B, int G, int E, int col0,
                                                      float* __restrict__ sim_v2t, float* __restrict__ sim_t2v, float* __restrict__ lse) {
  __shared__ float q[VTC_MAX_E];
  __shared__ float sh[4];
  const int dir = blockIdx.x / B, i = blockIdx.x - dir * B;
  const float* qrow = (dir == 0 ? v : t) + (int64_t)i * E;
  const float* keys = dir == 0 ? gt : gv;
  float* sim = (dir == 0 ? sim_v2t : sim_t2v) + (int64_t)i * G;
  const float inv_temp = 1.0f / fminf(fmaxf(*temp, 0.001f), 0.5f);  // temp.clamp_(0.001, 0.5), alpro_models.py:80-81
  for (int e = threadIdx.x; e < E; e += 256) q[e] = qrow[e];
  __syncthreads();
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < G; j += 256) {
    const float4* k4 = (const float4*)(keys + (int64_t)j * E);
    float acc = 0.f;
    for (int e = 0; e < E / 4; ++e) {
      const float4 k = k4[e];
      acc = fmaf(k.x, q[4 * e], fmaf(k.y, q[4 * e + 1], fmaf(k.z, q[4 * e + 2], fmaf(k.w, q[4 * e + 3], acc))));
    }
    acc *= inv_temp;
    sim[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = block_reduce(mx, sh, true);
  float s = 0.f;
  for (int j = threadIdx.x; j < G; j += 256) s += expf(sim[j] - mx);  // own writes: visible to the writing thread
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) lse[blockIdx.x] = mx + logf(s);   // the loss itself: vtc_loss_finish_kernel (fixed summation order)
}

// loss = sum over (direction, row) of (lse - positive logit) / (2B): one workgroup, thread k adds rows k, k + 256, ... in ascending order, then
// a fixed tree -- rounds 1-3 added the 2B terms with one fp32 atomic each, in whatever order the workgroups finished.
__global__ __launch_bounds__(256) void vtc_loss_finish_kernel(const float* __restrict__ sim_v2t, const float* __restrict__ sim_t2v,
                                                              const float* __restrict__ lse, int B, int G, int col0, float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int k = threadIdx.x; k < 2 * B; k += 256) {
    const int dir = k / B, i = k - dir * B;
    s += lse[k] - (dir == 0 ? sim_v2t : sim_t2v)[(int64_t)i * G + col0 + i];
  }
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) *loss = s * (0.5f / B);
}

// d(temp) = -sum over both directions of ds[i][j] * sim[i][j] / temp, same fixed order (the rows kernel used to add its row's share atomically)
__global__ __launch_bounds__(1024) void vtc_dtemp_kernel(const float* __restrict__ temp, const float* __restrict__ sim_v2t, const float* __restrict__ sim_t2v,
                                                         const float* __restrict__ ds_v2t, const float* __restrict__ ds_t2v, int64_t n,
                                                         float* __restrict__ dtemp) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int64_t k = threadIdx.x; k < n; k += 1024) s += ds_v2t[k] * sim_v2t[k];
  for (int64_t k = threadIdx.x; k < n; k += 1024) s += ds_t2v[k] * sim_t2v[k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sh[w];
    *dtemp = -t / fminf(fmaxf(*temp, 0.001f), 0.5f);
  }
}

// Backward, rows: ds[i][j] = (softmax(sim[i])[j] - [j == col0 + i]) * dloss / (2B); d(query i) = sum_j ds[i][j] key[j] / temp;
// d(temp) -= sum_j ds[i][j] sim[i][j] / temp.  ds is kept (B x G per direction) for the column pass.
__global__ __launch_bounds__(256) void vtc_bwd_rows_kernel(const float* __restrict__ gv, const float* __restrict__ gt, const float* __restrict__ temp,
                                                           int B, int G, int E, int col0, const float* __restrict__ sim_v2t,
                                                           const float* __restrict__ sim_t2v, const float* __restrict__ lse,
                                                           const float* __restrict__ dloss, float* __restrict__ ds_v2t, float* __restrict__ ds_t2v,
                                                           float* __restrict__ dv, float* __restrict__ dt) {
  extern __shared__ float dsrow[];  // G floats
  const int dir = blockIdx.x / B, i = blockIdx.x - dir * B;
  const float* keys = dir == 0 ? gt : gv;
  const float* sim = (dir == 0 ? sim_v2t : sim_t2v) + (int64_t)i * G;
  float* ds = (dir == 0 ? ds_v2t : ds_t2v) + (int64_t)i * G;
  const float tc = fminf(fmaxf(*temp, 0.001f), 0.5f);
  const float inv_temp = 1.0f / tc;
  const float scale = *dloss * (0.5f / B), l = lse[blockIdx.x];
  for (int j = threadIdx.x; j < G; j += 256) {
    const float d = (expf(sim[j] - l) - (j == col0 + i ? 1.f : 0.f)) * scale;
    dsrow[j] = d;
    ds[j] = d;
  }
  __syncthreads();  // publishes dsrow
  float* dq = (dir == 0 ? dv : dt) + (int64_t)i * E;
  for (int e = threadIdx.x; e < E; e += 256) {
    float acc = 0.f;
    for (int j = 0; j < G; ++j) acc = fmaf(dsrow[j], keys[(int64_t)j * E + e], acc);
    dq[e] = acc * inv_temp;
  }
}

// Backward, columns: d(key j of direction dir) = sum_i ds_dir[i][j] query_dir[i] / temp  (v2t: keys = gathered text, queries = v).
__global__ __launch_bounds__(256) void vtc_bwd_cols_kernel(const float* __restrict__ v, const float* __restrict__ t, const float* __restrict__ temp, int B,
                                                           int G, int E, const float* __restrict__ ds_v2t, const float* __restrict__ ds_t2v,
                                                           float* __restrict__ dgv, float* __restrict__ dgt) {
  const int dir = blockIdx.x / G, j = blockIdx.x - dir * G;
  const float* qs = dir == 0 ? v : t;
  const float* ds = dir == 0 ? ds_v2t : ds_t2v;
  float* dk = (dir == 0 ? dgt : dgv) + (int64_t)j * E;
  const float inv_temp = 1.0f / fminf(fmaxf(*temp, 0.001f), 0.5f);
  for (int e = threadIdx.x; e < E; e += 256) {
    float acc = 0.f;
    for (int i = 0; i < B; ++i) acc = fmaf(ds[(int64_t)i * G + j], qs[(int64_t)i * E + e], acc);
    dk[e] = acc * inv_temp;
  }
}
}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_vtc_loss_fwd(const float* v, const float* t, const float* gv, const float* gt, const float* temp, int B, int G, int E,
                                  int col0, float* sim_v2t, float* sim_t2v, float* lse, float* loss, void* stream) {
  ALPRO_CHECK(v && t && gv && gt && temp && sim_v2t && sim_t2v && lse && loss, "alpro_vtc_loss_fwd: null argument");
  ALPRO_CHECK(B > 0 && G >= B && E > 0 && E % 4 == 0 && E <= VTC_MAX_E && col0 >= 0 && col0 + B <= G, "alpro_vtc_loss_fwd: bad shape B=%d G=%d E=%d col0=%d", B, G, E, col0);
  hipLaunchKernelGGL(vtc_fwd_kernel, dim3(2 * B), dim3(256), 0, (hipStream_t)stream, v, t, gv, gt, temp, B, G, E, col0, sim_v2t, sim_t2v, lse);
  hipLaunchKernelGGL(vtc_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sim_v2t, sim_t2v, lse, B, G, col0, loss);
  return check_launch("alpro_vtc_loss_fwd");
}

extern "C" int alpro_vtc_loss_bwd(const float* v, const float* t, const float* gv, const float* gt, const float* temp, int B, int G, int E,
                                  int col0, const float* sim_v2t, const float* sim_t2v, const float* lse, const float* dloss, float* ds_v2t,
                                  float* ds_t2v, float* dv, float* dt, float* dgv, float* dgt, float* dtemp, void* stream) {
  ALPRO_CHECK(v && t && gv && gt && temp && sim_v2t && sim_t2v && lse && dloss && ds_v2t && ds_t2v && dv && dt && dgv && dgt, "alpro_vtc_loss_bwd: null argument");
  ALPRO_CHECK(B > 0 && G >= B && E > 0 && E % 4 == 0 && E <= VTC_MAX_E && G <= 8192, "alpro_vtc_loss_bwd: bad shape B=%d G=%d E=%d", B, G, E);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(vtc_bwd_rows_kernel, dim3(2 * B), dim3(256), G * sizeof(float), st, gv, gt, temp, B, G, E, col0, sim_v2t, sim_t2v, lse, dloss, ds_v2t,
                     ds_t2v, dv, dt);
  if (dtemp) hipLaunchKernelGGL(vtc_dtemp_kernel, dim3(1), dim3(1024), 0, st, temp, sim_v2t, sim_t2v, ds_v2t, ds_t2v, (int64_t)B * G, dtemp);
  hipLaunchKernelGGL(vtc_bwd_cols_kernel, dim3(2 * G), dim3(256), 0, st, v, t, temp, B, G, E, ds_v2t, ds_t2v, dgv, dgt);
  return check_launch("alpro_vtc_loss_bwd");
}

extern "C" int alpro_softmax_xent(const float* logits, int64_t ld, const int64_t* labels, int ignore_index, float* loss_rows, void* dlogits,
                                  int dtype, int64_t ldd, const float* grad_scale, int M, int V, int Vpad, void* stream) {
  ALPRO_CHECK(logits && labels && loss_rows && M > 0 && V > 0, "alpro_softmax_xent: bad args");
  ALPRO_CHECK(ld % 2 == 0 && ((uintptr_t)logits % 8) == 0, "alpro_softmax_xent: logits rows must be 8-byte aligned");
  ALPRO_CHECK(!dlogits || (grad_scale && Vpad >= V && Vpad % 2 == 0 && ldd >= Vpad && ldd % 2 == 0), "alpro_softmax_xent: bad gradient buffer");
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(xent_kernel<T>, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, ignore_index, loss_rows, (T*)dlogits, ldd, grad_scale, V, Vpad));
  return check_launch("alpro_softmax_xent");
}
