// Memory-bound backward kernels: transposes feeding the NT GEMM for dgrad / wgrad, LayerNorm backward
// (with the row maps of the forward gather turned into scatters), CLS-mean / final-pool / BERT-embedding
// backward, and the GELU-gradient multiply.  All HBM-bound; 16-byte accesses wherever rows are contiguous.
#include "common.hpp"
#include <algorithm>

namespace alpro {
namespace {

// ---- out[c, r] = (Tout) in[r, c] for r < R, zero for R <= r < Rpad; optional fp32 column sums (bias gradient) --
// 64x64 tile through LDS (fp32, +1 padding).  wgrad needs both operands with the token dimension contiguous.
template <typename Tin, typename Tout>
__device__ __forceinline__ void transpose_tile(float (&tile)[64][65], const Tin* __restrict__ in, int64_t ld_in, Tout* __restrict__ out,
                                               int64_t ld_out, int R, int C, int Rpad, float* __restrict__ colsum, int bx, int by) {
  const int r0 = by * 64, c0 = bx * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row-groups of 64 lanes
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? to_f32(in[(int64_t)r * ld_in + c]) : 0.f;
  }
  __syncthreads();
  if (colsum && ty == 0 && c0 + tx < C) {
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s += tile[i][tx];
    atomicAdd(colsum + c0 + tx, s);
  }
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) out[(int64_t)c * ld_out + r] = from_f32<Tout>(tile[tx][i]);
  }
}

template <typename Tin, typename Tout>
__global__ __launch_bounds__(256) void transpose_kernel(const Tin* __restrict__ in, int64_t ld_in, Tout* __restrict__ out, int64_t ld_out,
                                                        int R, int C, int Rpad, float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  transpose_tile<Tin, Tout>(tile, in, ld_in, out, ld_out, R, C, Rpad, colsum, blockIdx.x, blockIdx.y);
}

// Many independent transposes in ONE launch: the dgrad operands W^T of every Linear are refreshed after each optimizer step, ~130
// matrices of 0.6-2.4 M elements -- each far too small to cover its own launch (10 us apiece, 2 ms per step as separate launches).
// Workgroup -> job by binary search over the jobs' first-tile prefix (uniform, a handful of scalar loads).
template <typename Tout>
__global__ __launch_bounds__(256) void transpose_batch_kernel(const alpro_transpose_job_t* __restrict__ jobs, int njobs) {
  __shared__ float tile[64][65];
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const alpro_transpose_job_t jb = jobs[lo];
  const int t = blockIdx.x - jb.tile0, tcx = (jb.C + 63) / 64;
  transpose_tile<float, Tout>(tile, jb.in, jb.ld_in, (Tout*)jb.out, jb.ld_out, jb.R, jb.C, jb.Rpad, nullptr, t % tcx, t / tcx);
}

// ---- LayerNorm backward, D = 768, one wave per row ------------------------------------------------------
constexpr int LN_D = 768;
constexpr int LN_PART_BYTES = 3 * LN_D * (int)sizeof(float);  // one workgroup's column sums in the reduction workspace: dgamma | dbeta | colsum_pre

__device__ __forceinline__ void ld12(const float* row, int lane, float (&v)[12]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 f = *(const float4*)(row + i * 256 + lane * 4);
    v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
  }
}
// streamed-once rows (x, dy, the gradient stream): non-temporal
__device__ __forceinline__ void ld12_nt(const float* row, int lane, float (&v)[12]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const f32x4 f = __builtin_nontemporal_load((const f32x4*)(row + i * 256 + lane * 4));
    v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
  }
}
template <typename T> __device__ __forceinline__ void ld12_t(const T* row, int lane, float (&v)[12]) {
  if constexpr (sizeof(T) == 4) {
    ld12_nt((const float*)row, lane, v);
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const u32x2 u = __builtin_nontemporal_load((const u32x2*)(row + i * 256 + lane * 4));
      const uint32_t w0 = u.x, w1 = u.y;
      v[4 * i] = to_f32(T{(uint16_t)(w0 & 0xffffu)}); v[4 * i + 1] = to_f32(T{(uint16_t)(w0 >> 16)});
      v[4 * i + 2] = to_f32(T{(uint16_t)(w1 & 0xffffu)}); v[4 * i + 3] = to_f32(T{(uint16_t)(w1 >> 16)});
    }
  }
}

struct SrcRow {
  int64_t row;
  bool shared;  // the source row is gathered by several output rows (CLS under FRAME_TOKENS): scatter atomically
};
__device__ __forceinline__ SrcRow ln_src_row(int mode, int p0, int p1, int64_t m) {
  SrcRow s;
  s.shared = false;
  if (mode == ALPRO_MAP_IDENTITY) { s.row = m; return s; }
  if (mode == ALPRO_MAP_SKIP_CLS) { s.row = m + m / p0 + 1; return s; }
  const int T = p0, N = p1;
  const int64_t bt = m / (N + 1);
  const int j = (int)(m - bt * (N + 1));
  const int64_t b = bt / T;
  const int t = (int)(bt - b * T);
  const int64_t base = b * (1 + (int64_t)N * T);
  s.shared = j == 0;
  s.row = j == 0 ? base : base + 1 + (int64_t)(j - 1) * T + t;
  return s;
}

// What the backward does NEXT with the gradient row this kernel has just finished (round 3): every LayerNorm backward of the path is
// followed by an alpro_gather_cast that re-reads the fp32 row it wrote, scales it and casts it into the operand rows of the next
// wgrad / dgrad GEMMs (3.6 ms per training step, 308 MB re-read per call at B = 64).  With an emit mode the row leaves this kernel in
// both forms -- fp32 into the gradient stream and `dtype` into the GEMM operand -- and that pass disappears.  `r` = token row of dx:
//   ALPRO_EMIT_ROWS      out[r] = drop(v) * scale[r / group]                      (BERT: the dense-output dropout of xbert.py:358,436;
//                        ViT temporal LayerNorm -> the previous block's MLP gradient, scale = its drop-path row scale, group = 1 + N*T)
//   ALPRO_EMIT_FRAME     r = b*S + k.  k > 0 (patch n, frame t): out[(b*T+t)*(N+1) + 1 + n] = v * scale[b*T+t];
//                        k = 0: out[(b*T+t)*(N+1)] = v * scale[b*T+t] / T for every t   (inverse of vit.py:184-196; after norm2's backward)
//   ALPRO_EMIT_SKIP_CLS  k > 0: out[r - b - 1] = v * scale[(r - b - 1) / group]; colsum_pre[c] += v (unscaled: the bias gradient of
//                        temporal_fc); CLS rows emit nothing                      (after norm1's backward)
struct EmitArgs {
  void* out;
  int mode, p0, p1;          // p0 = T, p1 = N for the divided space-time modes
  const float* scale;
  int group;
  float drop_p;
  uint32_t drop_seed;
  float* colsum_pre;
  int extra_cls;             // ROWS mode after a SKIP_CLS-mapped LayerNorm: also emit the B CLS rows (final since the previous kernel)
};

template <typename T>
__device__ __forceinline__ void emit_store(T* p, int lane, const float (&v)[12], float sc) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T* q = p + i * 256 + lane * 4;
    if constexpr (sizeof(T) == 4) {
      *(f32x4*)q = f32x4{v[4 * i] * sc, v[4 * i + 1] * sc, v[4 * i + 2] * sc, v[4 * i + 3] * sc};
    } else {
      u32x2 u;
      u.x = pack2(v[4 * i] * sc, v[4 * i + 1] * sc, (T*)0);
      u.y = pack2(v[4 * i + 2] * sc, v[4 * i + 3] * sc, (T*)0);
      *(u32x2*)q = u;  // plain store: the wgrad / dgrad GEMMs read it next out of the Infinity Cache (like alpro_gather_cast)
    }
  }
}

// emit the finished gradient row `v` of token row r (see EmitArgs); cp accumulates the unscaled column sums (SKIP_CLS mode)
template <typename T>
__device__ __forceinline__ void emit_row(const EmitArgs& e, int64_t r, int lane, float (&v)[12], float (&cp)[12]) {
  T* out = (T*)e.out;
  if (e.mode == ALPRO_EMIT_ROWS) {
    if (e.drop_seed) {
      const uint32_t th = drop_thresh24(e.drop_p);
      const float ks = 1.0f / (1.0f - e.drop_p);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const uint64_t idx = (uint64_t)r * LN_D + (uint64_t)((i >> 2) * 256 + lane * 4 + (i & 3));
        v[i] = drop_keep(e.drop_seed, idx, th) ? v[i] * ks : 0.f;
      }
    }
    emit_store<T>(out + r * LN_D, lane, v, e.scale ? e.scale[r / e.group] : 1.0f);
    return;
  }
  const int Tn = e.p0, N = e.p1;
  const int64_t S = 1 + (int64_t)N * Tn;
  const int64_t b = r / S, k = r - b * S;
  if (e.mode == ALPRO_EMIT_FRAME) {
    if (k == 0) {
      const float inv = 1.0f / (float)Tn;
      for (int t = 0; t < Tn; ++t) emit_store<T>(out + ((b * Tn + t) * (N + 1)) * LN_D, lane, v, (e.scale ? e.scale[b * Tn + t] : 1.0f) * inv);
    } else {
      const int64_t n = (k - 1) / Tn;
      const int t = (int)((k - 1) - n * Tn);
      emit_store<T>(out + ((b * Tn + t) * (N + 1) + 1 + n) * LN_D, lane, v, e.scale ? e.scale[b * Tn + t] : 1.0f);
    }
  } else {  // SKIP_CLS
    if (k == 0) return;
    const int64_t o = r - b - 1;
#pragma unroll
    for (int i = 0; i < 12; ++i) cp[i] += v[i];
    emit_store<T>(out + o * LN_D, lane, v, e.scale ? e.scale[o / e.group] : 1.0f);
  }
}

// dx[src(m)] += rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat));  dgamma += dy*xhat;  dbeta += dy
// T = storage type of dy, TE = storage type of the emitted operand rows (the compute dtype; dy itself may be the fp32 stream)
template <typename T, typename TE>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, int64_t ld_dy, const float* __restrict__ dy2,
                                                            const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma, float eps,
                                                            float* __restrict__ dx, int64_t ld_dx, int accumulate, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int64_t rows, int mode, int p0, int p1, float drop_p,
                                                            uint32_t drop_seed, const EmitArgs em, float* __restrict__ part, float* __restrict__ cls_ws) {
  __shared__ float red[2][4][LN_D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float g[12], ag[12], ab[12], cp[12];
  ld12(gamma, lane, g);
#pragma unroll
  for (int i = 0; i < 12; ++i) ag[i] = ab[i] = cp[i] = 0.f;
  // one row: loads, statistics, dgamma / dbeta terms; fin = its input-gradient row (not stored here)
  auto row_grad = [&](int64_t m, int64_t srow, float (&fin)[12]) {
    float xv[12], d[12];
    ld12_nt(x + srow * ldx, lane, xv);
    ld12_t<T>(dy + m * ld_dy, lane, d);
    if (dy2) {  // second gradient stream on the same LN output (fp32 copy consumed as a residual)
      float d2[12];
      ld12_nt(dy2 + m * LN_D, lane, d2);
#pragma unroll
      for (int i = 0; i < 12; ++i) d[i] += d2[i];
    }
    if (drop_seed) {  // gradient through the dropout applied to this LayerNorm's output
      const uint32_t th = drop_thresh24(drop_p);
      const float ks = 1.0f / (1.0f - drop_p);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const uint64_t idx = (uint64_t)m * LN_D + (uint64_t)((i >> 2) * 256 + lane * 4 + (i & 3));
        d[i] = drop_keep(drop_seed, idx, th) ? d[i] * ks : 0.f;
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += xv[i];
    const float mean = wave_sum(s) * (1.0f / LN_D);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { xv[i] -= mean; qq += xv[i] * xv[i]; }
    const float rstd = rsqrtf(wave_sum(qq) * (1.0f / LN_D) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      xv[i] *= rstd;            // xhat
      ab[i] += d[i];
      ag[i] += d[i] * xv[i];
      d[i] *= g[i];             // dy * gamma
      s1 += d[i];
      s2 += d[i] * xv[i];
    }
    s1 = wave_sum(s1) * (1.0f / LN_D);
    s2 = wave_sum(s2) * (1.0f / LN_D);
#pragma unroll
    for (int i = 0; i < 12; ++i) fin[i] = rstd * (d[i] - s1 - xv[i] * s2);
  };
  for (int64_t m = wave; m < rows + em.extra_cls; m += nwaves) {
    if (m >= rows) {  // cast-only rows: the CLS rows a SKIP_CLS-mapped LayerNorm does not touch (their gradient is already final)
      const int64_t r = (m - rows) * (1 + (int64_t)em.p1 * em.p0);
      float v[12];
      ld12_nt(dx + r * ld_dx, lane, v);
      emit_row<TE>(em, r, lane, v, cp);
      continue;
    }
    const SrcRow src = ln_src_row(mode, p0, p1, m);
    // round 6: the row of dx that the result is added to is fetched WITH the row's x / dy (it used to be read behind the four dependent wave
    // reductions: a second exposed memory latency per row)
    f32x4 cin[3];
    const bool acc_here = accumulate && !src.shared;
    if (acc_here) {
      const float* o = dx + src.row * ld_dx;
#pragma unroll
      for (int i = 0; i < 3; ++i) cin[i] = __builtin_nontemporal_load((const f32x4*)(o + i * 256 + lane * 4));
    }
    float fin[12];  // the finished gradient row
    row_grad(m, src.row, fin);
    if (src.shared) {
      // FRAME_TOKENS: the clip's CLS row receives one term per frame.  With a workspace the term of frame copy m / (N + 1) = b * T + t is
      // parked in cls_ws[b * T + t] and cls_rows_reduce_kernel adds the T terms of a clip in frame order (one writer per row, fixed order);
      // without one, fp32 atomics straight into the row (rounds 1-3: run-to-run differences in the last bit)
      if (cls_ws) {
        float* o = cls_ws + (m / (p1 + 1)) * LN_D;
#pragma unroll
        for (int i = 0; i < 3; ++i) *(float4*)(o + i * 256 + lane * 4) = make_float4(fin[4 * i], fin[4 * i + 1], fin[4 * i + 2], fin[4 * i + 3]);
      } else {
        float* o = dx + src.row * ld_dx;
#pragma unroll
        for (int i = 0; i < 12; ++i) atomicAdd(o + (i >> 2) * 256 + lane * 4 + (i & 3), fin[i]);
      }
      continue;
    }
    float* o = dx + src.row * ld_dx;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float* p = o + i * 256 + lane * 4;
      if (acc_here) {
        fin[4 * i] += cin[i].x; fin[4 * i + 1] += cin[i].y; fin[4 * i + 2] += cin[i].z; fin[4 * i + 3] += cin[i].w;
      }
      __builtin_nontemporal_store(f32x4{fin[4 * i], fin[4 * i + 1], fin[4 * i + 2], fin[4 * i + 3]}, (f32x4*)p);
    }
    if (em.mode != ALPRO_EMIT_NONE) emit_row<TE>(em, src.row, lane, fin, cp);
  }
  // block reduction of dgamma / dbeta (and the emit's column sums).  part != nullptr: this workgroup's sums go to its slot of the caller's
  // workspace -- part[block][3][768] -- and colsum_reduce_kernel adds the slots in a fixed order (bit-reproducible, the default since
  // round 4); part == nullptr: one fp32 atomic per column per workgroup straight into the gradients (no workspace, order varies)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][w][i * 256 + lane * 4 + e] = ag[4 * i + e];
      red[1][w][i * 256 + lane * 4 + e] = ab[4 * i + e];
    }
  __syncthreads();
  float* slot = part ? part + (int64_t)blockIdx.x * (3 * LN_D) : nullptr;
  for (int c = threadIdx.x; c < LN_D; c += 256) {
    const float sg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    if (slot) { slot[c] = sg; slot[LN_D + c] = sb; }
    else { atomicAdd(dgamma + c, sg); atomicAdd(dbeta + c, sb); }
  }
  if (em.colsum_pre) {  // bias gradient of the Linear whose output gradient the emitted rows are (before the row scale)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[0][w][i * 256 + lane * 4 + e] = cp[4 * i + e];
    __syncthreads();
    for (int c = threadIdx.x; c < LN_D; c += 256) {
      const float sc = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
      if (slot) slot[2 * LN_D + c] = sc;
      else atomicAdd(em.colsum_pre + c, sc);
    }
  }
}

// dst_k[c] += sum over workgroup slots p of part[p][nslots][768] for the column sums k = 0 .. gridDim.y-1: ONE workgroup per (32 columns, k)
// -- 8 float4 column lanes x 128 slices of slots.  Slice s adds slots s, s + 128, ... in ascending order; then 8 x 8 lanes add 16 slice sums
// each (ascending) and one lane per column group adds those 8: a single writer per output and an order that depends on nothing but the slot
// count -- bit-reproducible.  (72 workgroups; the first form -- 36 workgroups of 64 columns, one lane walking all 64 slice sums -- took 13 us
// for 2048 slots.)
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ part, int nparts, int nslots, float* __restrict__ d0,
                                                             float* __restrict__ d1, float* __restrict__ d2) {
  __shared__ float4 red[128][8];
  __shared__ float4 red2[8][8];
  const int k = blockIdx.y, c4 = threadIdx.x & 7, sl = threadIdx.x >> 3;
  float* dst = k == 0 ? d0 : (k == 1 ? d1 : d2);
  if (!dst) return;
  const int col = blockIdx.x * 32 + c4 * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int p = sl; p < nparts; p += 128) {
    const float4 v = *(const float4*)(part + ((int64_t)p * nslots + k) * LN_D + col);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  red[sl][c4] = s;
  __syncthreads();
  if (sl < 8) {
    float4 t = red[sl * 16][c4];
#pragma unroll
    for (int i = 1; i < 16; ++i) { const float4 v = red[sl * 16 + i][c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    red2[sl][c4] = t;
  }
  __syncthreads();
  if (sl == 0) {
    float4 t = red2[0][c4];
#pragma unroll
    for (int i = 1; i < 8; ++i) { const float4 v = red2[i][c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    float4* o = (float4*)(dst + col);
    const float4 c = *o;
    *o = make_float4(c.x + t.x, c.y + t.y, c.z + t.z, c.w + t.w);
  }
}

// dx[clip b's CLS row] += sum over the T frame terms parked in cls_ws[b * T + t] by layernorm_bwd_kernel, t ascending: one workgroup of 192
// lanes (float4 each) per clip.
__global__ __launch_bounds__(192) void cls_rows_reduce_kernel(const float* __restrict__ cls_ws, float* __restrict__ dx, int64_t ld_dx, int T, int N) {
  const int b = blockIdx.x, c = threadIdx.x * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = 0; t < T; ++t) {
    const float4 v = *(const float4*)(cls_ws + ((int64_t)b * T + t) * LN_D + c);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float4* o = (float4*)(dx + (int64_t)b * (1 + (int64_t)N * T) * ld_dx + c);
  const float4 cur = *o;
  *o = make_float4(cur.x + s.x, cur.y + s.y, cur.z + s.z, cur.w + s.w);
}

// ---- out[m, :] = (T)(scale(m) * src[row(m), :]): gathers fp32 token-gradient rows into a GEMM operand ---------------
// row(m) follows the forward maps; under FRAME_TOKENS the j == 0 rows read the clip's CLS row times cls_scale (1/T).
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void gather_cast_kernel(const float* __restrict__ src, int64_t ld, T* __restrict__ out, int64_t rows, int mode,
                                                          int p0, int p1, const float* __restrict__ row_scale, int group, float cls_scale,
                                                          float drop_p, uint32_t drop_seed, float* __restrict__ colsum,
                                                          float* __restrict__ colsum_pre, float* __restrict__ part) {
  __shared__ float red[NW][LN_D];
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  float cs[12], cp[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) cs[i] = cp[i] = 0.f;
  auto src_row = [&](int64_t m, float& sc) -> int64_t {
    sc = row_scale ? row_scale[m / group] : 1.0f;
    if (mode == ALPRO_MAP_PATCH_EMBED) {
      const int T_ = p0, N = p1;
      const int64_t bt = m / N;
      const int n = (int)(m - bt * N);
      const int64_t b = bt / T_;
      const int t = (int)(bt - b * T_);
      return b * (1 + (int64_t)N * T_) + 1 + (int64_t)n * T_ + t;
    }
    const SrcRow s = ln_src_row(mode, p0, p1, m);
    if (s.shared) sc *= cls_scale;
    return s.row;
  };
  auto emit = [&](int64_t m, float (&v)[12], float sc) {
    if (drop_seed) {
      const uint32_t th = drop_thresh24(drop_p);
      const float ks = 1.0f / (1.0f - drop_p);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const uint64_t idx = (uint64_t)m * LN_D + (uint64_t)((i >> 2) * 256 + lane * 4 + (i & 3));
        v[i] = drop_keep(drop_seed, idx, th) ? v[i] * ks : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      cp[i] += v[i];
      v[i] *= sc;
      cs[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      T* p = out + m * LN_D + i * 256 + lane * 4;
      if constexpr (sizeof(T) == 4) {
        __builtin_nontemporal_store(f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}, (f32x4*)p);
      } else {
        u32x2 u;
        u.x = pack2(v[4 * i], v[4 * i + 1], (T*)0);
        u.y = pack2(v[4 * i + 2], v[4 * i + 3], (T*)0);
        *(u32x2*)p = u;  // plain store: the wgrad / dgrad GEMMs read it next out of the Infinity Cache (see ln_store in core.hip)
      }
    }
  };
  // two rows per wave iteration: both rows' loads are in flight before either is converted (the colsum launch uses a
  // small grid, so each wave walks ~25 rows and would otherwise expose one HBM round trip per row)
  for (int64_t m = wave; m < rows; m += 2 * nwaves) {
    const int64_t m2 = m + nwaves;
    const bool has2 = m2 < rows;
    float sc, sc2 = 0.f;
    const int64_t r = src_row(m, sc);
    const int64_t r2 = has2 ? src_row(m2, sc2) : r;
    float v[12], v2[12];
    ld12_nt(src + r * ld, lane, v);
    ld12_nt(src + r2 * ld, lane, v2);
    emit(m, v, sc);
    if (has2) emit(m2, v2, sc2);
  }
  // bias gradients of the Linears these rows are the dY of (after / before the row scale): LDS fold over the waves, then this workgroup's
  // sums go to its workspace slot part[block][2][768] for colsum_reduce_kernel (fixed order), or -- part == nullptr -- one atomic per column
  auto flush = [&](const float (&acc)[12], float* dst, int k) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[w][i * 256 + lane * 4 + e] = acc[4 * i + e];
    __syncthreads();
    for (int c = threadIdx.x; c < LN_D; c += NW * 64) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < NW; ++j) t += red[j][c];
      if (part) part[((int64_t)blockIdx.x * 2 + k) * LN_D + c] = t;
      else unsafeAtomicAdd(dst + c, t);
    }
  };
  if (colsum) flush(cs, colsum, 0);
  if (colsum_pre) flush(cp, colsum_pre, 1);
}

// ---- du = dh * gelu'(u) (erf GELU), elementwise over 16-byte chunks ---------------------------------------
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ u, T* __restrict__ du, int64_t n) {
  constexpr int E = Chunk<T>::N;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * E;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * E; i + E <= n; i += stride) {
    float a[E], b[E], o[E];
    unpack_chunk<T>(*(const u32x4*)(dh + i), a);
    unpack_chunk<T>(*(const u32x4*)(u + i), b);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float x = b[e];
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
      o[e] = a[e] * (cdf + x * pdf);
    }
    *(u32x4*)(du + i) = pack_chunk<T>(o);
  }
}

// ---- CLS mean backward: dside[b*T+t] = dx_out[b,0] / T  (rows of the spatial proj output), dx_in[b,0] += dx_out[b,0]
__global__ void cls_mean_bwd_kernel(const float* __restrict__ dx_out, int64_t ldb, float* __restrict__ dside, int B, int T, int D) {
  const int b = blockIdx.x;
  const float inv = 1.0f / (float)T;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float g = dx_out[b * ldb + d] * inv;
    for (int t = 0; t < T; ++t) dside[((int64_t)b * T + t) * D + d] = g;
  }
}

// ---- dst[idx[i], :] += src[i, :] (word / position embedding gradients), D = 768 ---------------------------
// Rows whose index equals skip_idx contribute nothing: nn.Embedding(padding_idx=...) keeps the pad row's gradient at zero (xbert.py:171).
// fp32 atomics: the order of the duplicates' additions varies run to run (the bit-reproducible caller sorts instead, see hip.py).
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst,
                                                               int rows, int idx_mod, int64_t skip_idx) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int64_t r = idx ? idx[m] : (m % idx_mod);
  if (r == skip_idx) return;
  const float* s = src + (int64_t)m * LN_D;
  float* d = dst + r * LN_D;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 f = *(const float4*)(s + i * 256 + lane * 4);
    float* p = d + i * 256 + lane * 4;
    atomicAdd(p, f.x); atomicAdd(p + 1, f.y); atomicAdd(p + 2, f.z); atomicAdd(p + 3, f.w);
  }
}

// idx == nullptr (destination row = m % idx_mod, the position table): one wave per DESTINATION row adds its rows m = r, r + idx_mod, ...
// in ascending order -- single writer, fixed order, no atomics.
__global__ __launch_bounds__(256) void scatter_add_mod_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int idx_mod) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= idx_mod || r >= rows) return;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int m = r; m < rows; m += idx_mod) {
    float v[12];
    ld12_nt(src + (int64_t)m * LN_D, lane, v);
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] += v[i];
  }
  float* d = dst + (int64_t)r * LN_D;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float4* p = (float4*)(d + i * 256 + lane * 4);
    const float4 c = *p;
    *p = make_float4(c.x + acc[4 * i], c.y + acc[4 * i + 1], c.z + acc[4 * i + 2], c.w + acc[4 * i + 3]);
  }
}

// ---- indexed scatter in a fixed order (round 5): the word-embedding table's gradient (xbert.py:203-210 backward) ---------------------------
// dst[idx[i]] += src[i] with duplicates ([CLS], [SEP], [MASK] appear hundreds of times) added in ascending i, one writer per destination row.
// Rounds 1-4 went through torch's sort-based index_put_ for that (0.36 ms per step of generic sort / gather / scatter kernels).  Two launches
// here: (1) ONE workgroup sorts the composite keys (idx << 13 | i) of up to 8192 rows in LDS (bitonic network, 32 KiB); (2) one wave per
// sorted position: the head of a run of equal destinations sums the run's rows in key order (= ascending i) and adds the total to the table.
constexpr int SCATTER_SORT_MAX = 8192;
__global__ __launch_bounds__(1024) void scatter_sort_keys_kernel(const int64_t* __restrict__ idx, int rows, int64_t dst_rows, uint32_t* __restrict__ keys_out) {
  __shared__ uint32_t k[SCATTER_SORT_MAX];
  // An index outside [0, dst_rows) gets the padding key: its row is DROPPED (sorted behind every valid key, skipped by the run kernel).  Truncated
  // into the 19-bit field it would have landed in somebody else's row -- or beyond the table (torch's index_put_ device-asserts there).
  for (int i = threadIdx.x; i < SCATTER_SORT_MAX; i += 1024) {
    const int64_t d = i < rows ? idx[i] : -1;
    k[i] = (d >= 0 && d < dst_rows) ? (((uint32_t)d << 13) | (uint32_t)i) : 0xFFFFFFFFu;
  }
  __syncthreads();
  for (int size = 2; size <= SCATTER_SORT_MAX; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < SCATTER_SORT_MAX / 2; t += 1024) {
        const int lo = 2 * t - (t & (stride - 1));   // the lower element of the t-th compare-exchange pair at this stride
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint32_t a = k[lo], b = k[hi];
        if ((a > b) == up) {
          k[lo] = b;
          k[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < SCATTER_SORT_MAX; i += 1024) keys_out[i] = k[i];
}

__global__ __launch_bounds__(256) void scatter_add_runs_kernel(const float* __restrict__ src, const uint32_t* __restrict__ keys, float* __restrict__ dst, int rows,
                                                               int64_t skip_idx) {
  const int lane = threadIdx.x & 63;
  const int pos = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= rows) return;
  const uint32_t key = keys[pos];
  if (key == 0xFFFFFFFFu) return;   // padding / an out-of-range index (dst_rows < 2^19: no valid key has this value)
  const uint32_t dest = key >> 13;
  if ((pos > 0 && (keys[pos - 1] >> 13) == dest) || (int64_t)dest == skip_idx) return;   // not the head of its run / the padding row (nn.Embedding(padding_idx))
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int q = pos; q < rows; ++q) {
    const uint32_t kq = keys[q];
    if ((kq >> 13) != dest) break;
    float v[12];
    ld12_nt(src + (int64_t)(kq & 8191u) * LN_D, lane, v);
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] += v[i];
  }
  float* d = dst + (int64_t)dest * LN_D;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float4* p = (float4*)(d + i * 256 + lane * 4);
    const float4 c = *p;
    *p = make_float4(c.x + acc[4 * i], c.y + acc[4 * i + 1], c.z + acc[4 * i + 2], c.w + acc[4 * i + 3]);
  }
}

// ---- small per-block terms of the merged temporal projection W_e = W_fc W_p (vit.py:157-162), all ViT blocks in one launch -------------
//   mode 0 (after an optimizer step):  b1[n] = sum_m W_fc[n, m] b_p[m]                      (the merged bias under the drop-path scale)
//   mode 1 (backward, product rule):   g_fc[n, m] += db1[n] b_p[m];   g_bp[m] += sum_n W_fc[n, m] db1[n]
// One workgroup per (job, 64-column block m0..m0+63) in mode 1 -- a single writer per output element and a fixed summation order, so the
// gradients stay bit-reproducible --, one wave per row n in mode 0.  D = 768.
template <int MODE>
__global__ __launch_bounds__(256) void tproj_small_kernel(const alpro_tproj_job_t* __restrict__ jobs, int D) {
  const alpro_tproj_job_t jb = jobs[blockIdx.y];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (MODE == 0) {
    const int n = blockIdx.x * 4 + w;
    if (n >= D) return;
    const float* row = jb.wfc + (int64_t)n * D;
    float acc = 0.f;
    for (int m = lane * 4; m < D; m += 256) {
      const float4 a = *(const float4*)(row + m), b = *(const float4*)(jb.bp + m);
      acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) jb.b1[n] = acc;
  } else {
    __shared__ float red[4][64];
    const int m = blockIdx.x * 64 + lane;
    const float bpm = jb.bp[m];
    // eight rows per trip, all 24 loads issued before the first store (round 6: one row per trip was 192 dependent memory round trips per wave,
    // 150 us for a launch that moves 20 MB); the sum over n keeps a fixed order (rows w, w + 4, ... in ascending order)
    float acc = 0.f;
    for (int n0 = w; n0 < D; n0 += 32) {   // D = 768: 24 trips of 8 rows (n0 + 4 k < D for every k)
      float d[8], gf[8], wf[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int n = n0 + 4 * k;
        const int64_t i = (int64_t)n * D + m;
        d[k] = jb.db1[n];
        gf[k] = jb.g_fc[i];
        wf[k] = jb.wfc[i];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        jb.g_fc[(int64_t)(n0 + 4 * k) * D + m] = gf[k] + d[k] * bpm;
        acc = fmaf(wf[k], d[k], acc);
      }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0) jb.g_bp[m] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  }
}

inline int grid_for(int64_t work_items, int per_block, int cap) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

template <typename Tin, typename Tout>
int launch_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int R, int C, int Rpad, float* colsum, hipStream_t st) {
  dim3 grid((C + 63) / 64, (Rpad + 63) / 64);
  hipLaunchKernelGGL((transpose_kernel<Tin, Tout>), grid, dim3(256), 0, st, (const Tin*)in, ld_in, (Tout*)out, ld_out, R, C, Rpad, colsum);
  return check_launch("alpro_transpose");
}

}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_transpose(const void* in, int in_dtype, int64_t ld_in, void* out, int out_dtype, int64_t ld_out, int R, int C,
                               int Rpad, float* colsum, void* stream) {
  ALPRO_CHECK(in && out && R > 0 && C > 0 && Rpad >= R, "alpro_transpose: bad args");
  ALPRO_CHECK(ld_out >= Rpad && ld_in >= C, "alpro_transpose: leading dimensions too small");
  ALPRO_CHECK(in_dtype == out_dtype || in_dtype == ALPRO_F32, "alpro_transpose: input must be fp32 or the output dtype");
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == ALPRO_F32) {
    ALPRO_DISPATCH_DTYPE(out_dtype, T, return (launch_transpose<float, T>(in, ld_in, out, ld_out, R, C, Rpad, colsum, st)));
  } else {
    ALPRO_DISPATCH_DTYPE(out_dtype, T, return (launch_transpose<T, T>(in, ld_in, out, ld_out, R, C, Rpad, colsum, st)));
  }
  return ALPRO_OK;
}

extern "C" int alpro_transpose_batch(const alpro_transpose_job_t* jobs, int njobs, int total_tiles, int out_dtype, void* stream) {
  ALPRO_CHECK(jobs && njobs > 0 && total_tiles > 0, "alpro_transpose_batch: bad args");
  ALPRO_DISPATCH_DTYPE(out_dtype, T, hipLaunchKernelGGL(transpose_batch_kernel<T>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, jobs, njobs));
  return check_launch("alpro_transpose_batch");
}

extern "C" int alpro_layernorm_bwd_emit(const void* dy, int dy_dtype, int64_t ld_dy, const float* dy2, const float* x, int64_t ldx,
                                        const float* gamma, float eps, float* dx, int64_t ld_dx, int accumulate, float* dgamma, float* dbeta,
                                        int rows, int D, int map_mode, int map_p0, int map_p1, float drop_p, uint32_t drop_seed, void* emit_out,
                                        int emit_dtype, int emit_mode, int emit_p0, int emit_p1, const float* emit_scale, int emit_scale_group, float emit_drop_p,
                                        uint32_t emit_drop_seed, float* emit_colsum_pre, int emit_extra_cls, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  ALPRO_CHECK(dy && x && gamma && dx && dgamma && dbeta && rows > 0, "alpro_layernorm_bwd: bad args");
  ALPRO_CHECK(!workspace || (((uintptr_t)workspace % 16) == 0 && workspace_bytes >= LN_PART_BYTES), "alpro_layernorm_bwd: the workspace must be 16-byte aligned and hold at least one workgroup's sums (%d bytes)", LN_PART_BYTES);
  ALPRO_CHECK(D == LN_D, "alpro_layernorm_bwd: D=%d unsupported", D);
  ALPRO_CHECK(map_mode >= 0 && map_mode <= ALPRO_MAP_FRAME_TOKENS, "alpro_layernorm_bwd: bad map_mode %d", map_mode);
  ALPRO_CHECK(map_mode != ALPRO_MAP_FRAME_TOKENS || accumulate, "alpro_layernorm_bwd: the FRAME_TOKENS scatter needs accumulate=1 (CLS rows are shared)");
  ALPRO_CHECK(emit_mode >= ALPRO_EMIT_NONE && emit_mode <= ALPRO_EMIT_SKIP_CLS, "alpro_layernorm_bwd_emit: bad emit_mode %d", emit_mode);
  ALPRO_CHECK(emit_mode == ALPRO_EMIT_NONE || emit_out, "alpro_layernorm_bwd_emit: an emit mode needs the output rows");
  ALPRO_CHECK(emit_mode == ALPRO_EMIT_NONE || ld_dx == LN_D, "alpro_layernorm_bwd_emit: the emit needs a dense gradient stream (ld_dx == 768)");
  ALPRO_CHECK(emit_mode == ALPRO_EMIT_NONE || emit_mode == ALPRO_EMIT_ROWS || (emit_p0 > 0 && emit_p1 > 0), "alpro_layernorm_bwd_emit: FRAME / SKIP_CLS emits need p0 = T, p1 = N");
  ALPRO_CHECK(!emit_scale || emit_scale_group > 0, "alpro_layernorm_bwd_emit: emit_scale_group must be > 0");
  ALPRO_CHECK(!emit_drop_seed || (emit_mode == ALPRO_EMIT_ROWS && emit_drop_p > 0.f && emit_drop_p < 1.f), "alpro_layernorm_bwd_emit: dropout only in ROWS mode, 0 < p < 1");
  ALPRO_CHECK(!emit_colsum_pre || emit_mode == ALPRO_EMIT_SKIP_CLS, "alpro_layernorm_bwd_emit: colsum_pre belongs to the SKIP_CLS emit");
  ALPRO_CHECK(emit_extra_cls == 0 || (emit_mode == ALPRO_EMIT_ROWS && map_mode == ALPRO_MAP_SKIP_CLS && emit_p0 > 0 && emit_p1 > 0),
              "alpro_layernorm_bwd_emit: extra CLS rows go with ROWS emit after a SKIP_CLS-mapped LayerNorm (p0 = T, p1 = N)");
  EmitArgs em;
  em.out = emit_out; em.mode = emit_mode; em.p0 = emit_p0; em.p1 = emit_p1; em.scale = emit_scale; em.group = emit_scale_group;
  em.drop_p = emit_drop_p; em.drop_seed = emit_drop_seed; em.colsum_pre = emit_colsum_pre; em.extra_cls = emit_extra_cls;
  ALPRO_CHECK(emit_mode == ALPRO_EMIT_NONE || emit_dtype == dy_dtype || dy_dtype == ALPRO_F32, "alpro_layernorm_bwd_emit: emit_dtype must be dy's dtype, or any dtype when dy is the fp32 stream");
  // workspace given: the column sums (dgamma, dbeta, colsum_pre) leave as per-workgroup partials and are added in a fixed order by a second
  // small kernel, and the T frame terms of every clip's CLS row (FRAME_TOKENS) are parked behind them for cls_rows_reduce_kernel --
  // bit-reproducible.  Layout: [workgroups][3][768] floats, then [B * T][768] floats; the grid is capped to what fits.  NULL: fp32 atomics.
  const bool frame = map_mode == ALPRO_MAP_FRAME_TOKENS;
  const size_t cls_bytes = frame ? (size_t)(rows / (map_p1 + 1)) * LN_D * sizeof(float) : 0;
  ALPRO_CHECK(!workspace || workspace_bytes >= cls_bytes + LN_PART_BYTES, "alpro_layernorm_bwd: the workspace must hold the CLS frame terms (%zu bytes) and at least one workgroup's sums", cls_bytes);
  // workgroup count: 8 rows per wave at least, 2048 workgroups at most -- two 4-wave workgroups per CU are resident at a time (185 VGPRs), so
  // that is 4 rounds; fewer, fatter workgroups measured slower (tools/ln_bwd_bench.py: 1024: +14 %, 512: +5 %; `ln_grid` overrides the cap)
  const int cap = get_option(OPT_LN_GRID);
  int nblk = grid_for(rows, 4 * 8, cap > 0 ? cap : 256 * 8);
  if (workspace) nblk = (int)std::min<size_t>((size_t)nblk, (workspace_bytes - cls_bytes) / LN_PART_BYTES);
  const dim3 grid(nblk), blk(256);
  float* part = (float*)workspace;
  float* cls_ws = (workspace && frame) ? part + (size_t)nblk * 3 * LN_D : nullptr;
  hipStream_t st = (hipStream_t)stream;
#define ALPRO_LNB(T, TE) hipLaunchKernelGGL((layernorm_bwd_kernel<T, TE>), grid, blk, 0, st, (const T*)dy, ld_dy, dy2, x, ldx, gamma, eps, dx, ld_dx, accumulate, dgamma, dbeta, (int64_t)rows, map_mode, map_p0, map_p1, drop_p, drop_seed, em, part, cls_ws)
  if (dy_dtype == ALPRO_F32 && emit_mode != ALPRO_EMIT_NONE && emit_dtype == ALPRO_BF16) ALPRO_LNB(float, bf16_t);
  else if (dy_dtype == ALPRO_F32 && emit_mode != ALPRO_EMIT_NONE && emit_dtype == ALPRO_F16) ALPRO_LNB(float, f16_t);
  else if (dy_dtype == ALPRO_F32) ALPRO_LNB(float, float);
  else if (dy_dtype == ALPRO_BF16) ALPRO_LNB(bf16_t, bf16_t);
  else if (dy_dtype == ALPRO_F16) ALPRO_LNB(f16_t, f16_t);
  else { set_error("alpro_layernorm_bwd: bad dtype %d", dy_dtype); return ALPRO_ERR_INVALID; }
#undef ALPRO_LNB
  if (part) hipLaunchKernelGGL(colsum_reduce_kernel, dim3(LN_D / 32, emit_colsum_pre ? 3 : 2), dim3(1024), 0, st, part, nblk, 3, dgamma, dbeta, emit_colsum_pre);
  if (cls_ws) hipLaunchKernelGGL(cls_rows_reduce_kernel, dim3(rows / (map_p1 + 1) / map_p0), dim3(192), 0, st, cls_ws, dx, ld_dx, map_p0, map_p1);
  return check_launch("alpro_layernorm_bwd");
}

extern "C" int alpro_layernorm_bwd(const void* dy, int dy_dtype, int64_t ld_dy, const float* dy2, const float* x, int64_t ldx,
                                   const float* gamma, float eps, float* dx, int64_t ld_dx, int accumulate, float* dgamma, float* dbeta,
                                   int rows, int D, int map_mode, int map_p0, int map_p1, float drop_p, uint32_t drop_seed, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return alpro_layernorm_bwd_emit(dy, dy_dtype, ld_dy, dy2, x, ldx, gamma, eps, dx, ld_dx, accumulate, dgamma, dbeta, rows, D, map_mode, map_p0, map_p1,
                                  drop_p, drop_seed, nullptr, dy_dtype, ALPRO_EMIT_NONE, 0, 0, nullptr, 1, 0.f, 0u, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int alpro_gather_cast(const float* src, int64_t ld, void* out, int dtype, int rows, int D, int map_mode, int map_p0, int map_p1,
                                 const float* row_scale, int row_scale_group, float cls_scale, float drop_p, uint32_t drop_seed, float* colsum,
                                 float* colsum_pre, void* workspace, size_t workspace_bytes, void* stream) {
  ALPRO_CHECK(src && out && rows > 0, "alpro_gather_cast: bad args");
  ALPRO_CHECK(!workspace || (((uintptr_t)workspace % 16) == 0 && workspace_bytes >= 2 * LN_D * sizeof(float)), "alpro_gather_cast: the workspace must be 16-byte aligned and hold at least one workgroup's sums");
  ALPRO_CHECK(D == LN_D, "alpro_gather_cast: D=%d unsupported", D);
  ALPRO_CHECK(map_mode >= 0 && map_mode <= ALPRO_MAP_PATCH_EMBED, "alpro_gather_cast: bad map_mode %d", map_mode);
  ALPRO_CHECK(!row_scale || row_scale_group > 0, "alpro_gather_cast: row_scale_group must be > 0");
  // with a colsum target: few, fat workgroups (16 waves) -- every workgroup ends with 768 same-address atomics, and those
  // serialise at the memory side (3072 workgroups cost 2x the whole cast); plain casts use many small workgroups
  if (colsum || colsum_pre) {
    // workspace given: per-workgroup partial sums + the fixed-order colsum_reduce_kernel (bit-reproducible); NULL: fp32 atomics
    float* part = (float*)workspace;
    int nblk = grid_for(rows, 32, 512);
    if (part) nblk = (int)std::min<size_t>((size_t)nblk, workspace_bytes / (2 * LN_D * sizeof(float)));
    ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gather_cast_kernel<T, 16>), dim3(nblk), dim3(1024), 0, (hipStream_t)stream, src, ld, (T*)out, (int64_t)rows, map_mode, map_p0, map_p1, row_scale, row_scale_group, cls_scale, drop_p, drop_seed, colsum, colsum_pre, part));
    if (part) hipLaunchKernelGGL(colsum_reduce_kernel, dim3(LN_D / 32, 2), dim3(1024), 0, (hipStream_t)stream, part, nblk, 2, colsum, colsum_pre, (float*)nullptr);
  } else {
    ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gather_cast_kernel<T, 4>), dim3(grid_for(rows, 8, 256 * 32)), dim3(256), 0, (hipStream_t)stream, src, ld, (T*)out, (int64_t)rows, map_mode, map_p0, map_p1, row_scale, row_scale_group, cls_scale, drop_p, drop_seed, colsum, colsum_pre, (float*)nullptr));
  }
  return check_launch("alpro_gather_cast");
}

extern "C" int alpro_tproj_small(const alpro_tproj_job_t* jobs_device, int njobs, int D, int mode, void* stream) {
  ALPRO_CHECK(jobs_device && njobs > 0 && (mode == 0 || mode == 1), "alpro_tproj_small: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_tproj_small: D=%d unsupported", D);
  if (mode == 0) hipLaunchKernelGGL(tproj_small_kernel<0>, dim3(D / 4, njobs), dim3(256), 0, (hipStream_t)stream, jobs_device, D);
  else hipLaunchKernelGGL(tproj_small_kernel<1>, dim3(D / 64, njobs), dim3(256), 0, (hipStream_t)stream, jobs_device, D);
  return check_launch("alpro_tproj_small");
}

extern "C" int alpro_gelu_bwd(const void* dh, const void* u, void* du, int dtype, int64_t n, void* stream) {
  ALPRO_CHECK(dh && u && du && n > 0, "alpro_gelu_bwd: bad args");
  ALPRO_CHECK(n % 8 == 0, "alpro_gelu_bwd: n must be a multiple of 8");
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(grid_for(n / Chunk<T>::N, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (const T*)dh, (const T*)u, (T*)du, n));
  return check_launch("alpro_gelu_bwd");
}

extern "C" int alpro_cls_mean_bwd(const float* dx_out, int64_t ld_batch, float* dside, int B, int T, int D, void* stream) {
  ALPRO_CHECK(dx_out && dside && B > 0 && T > 0 && D > 0, "alpro_cls_mean_bwd: bad args");
  hipLaunchKernelGGL(cls_mean_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dx_out, ld_batch, dside, B, T, D);
  return check_launch("alpro_cls_mean_bwd");
}

extern "C" int alpro_scatter_add_rows_ordered(const float* src, const int64_t* idx, float* dst, int rows, int D, int64_t dst_rows, int64_t skip_idx, uint32_t* keys_ws,
                                              void* stream) {
  ALPRO_CHECK(src && idx && dst && keys_ws && rows > 0, "alpro_scatter_add_rows_ordered: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_scatter_add_rows_ordered: D=%d unsupported", D);
  ALPRO_CHECK(rows <= SCATTER_SORT_MAX && dst_rows > 0 && dst_rows < (1 << 19), "alpro_scatter_add_rows_ordered: at most %d source rows and fewer than 2^19 table rows (got %d, %lld)",
              SCATTER_SORT_MAX, rows, (long long)dst_rows);
  hipLaunchKernelGGL(scatter_sort_keys_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, idx, rows, dst_rows, keys_ws);
  hipLaunchKernelGGL(scatter_add_runs_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, keys_ws, dst, rows, skip_idx);
  return check_launch("alpro_scatter_add_rows_ordered");
}

extern "C" int alpro_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int rows, int idx_mod, int D, int64_t skip_idx, void* stream) {
  ALPRO_CHECK(src && dst && rows > 0 && (idx || idx_mod > 0), "alpro_scatter_add_rows: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_scatter_add_rows: D=%d unsupported", D);
  if (!idx && skip_idx < 0) {  // destination = row % idx_mod: one wave per destination row, ascending order (bit-reproducible)
    hipLaunchKernelGGL(scatter_add_mod_kernel, dim3((idx_mod + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, dst, rows, idx_mod);
    return check_launch("alpro_scatter_add_rows");
  }
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, idx, dst, rows, idx_mod, skip_idx);
  return check_launch("alpro_scatter_add_rows");
}
