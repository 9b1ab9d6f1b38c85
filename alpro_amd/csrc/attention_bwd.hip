// Attention backward for gfx950 (head_dim 64): dQ, dK, dV from dO with the softmax recomputed from the saved
// row log-sum-exp.  Same MFMA formulation as the forward (attention.hip): every product keeps its contraction
// index in registers by choosing the orientation per product, so no accumulator ever crosses lanes.
//
//   phase 1 (K, V tiles resident in LDS; one 32-query tile per wave; lane = query)
//       S^T = K Q^T, dP^T = V dO^T                  A = rows of K / V (ds_read_b128), B = Q / dO rows (registers)
//       P^T = exp(S^T*scale + bias - lse[q]),  dS^T = P^T o (dP^T - delta[q]) * scale,  delta = rowsum(dO o O)
//       dQ^T = K^T dS^T                             A = K^T via ds_read_b64_tr_b16, B = dS^T straight from registers
//   phase 2 (Q, dO tiles re-staged into the same LDS; one 32-key tile per wave; lane = key)
//       S = Q K^T, dP = dO V^T                      A = rows of Q / dO, B = K / V rows of the key tile (registers)
//       dV^T += dO^T P,  dK^T += Q^T dS             A = dO^T / Q^T via transpose reads, B = P / dS from registers
// One LDS swizzle serves both access kinds for 16-bit tiles: chunk ^= (bit1(row) << 2 | (row >> 2) & 3) is a bijection
// of (row >> 1) & 7 (conflict-free ds_read_b128 fragments) AND moves rows r, r+2 into different 64-byte windows
// (conflict-free 4-row transpose gathers).  The temporal variant treats 32 consecutive tokens as one tile with a
// block-diagonal group mask (one wave per workgroup).
#include "common.hpp"

namespace alpro {
namespace {

constexpr int HD = 64;

template <typename T> struct BCfg {
  static constexpr int E = sizeof(T);
  static constexpr int CN = 16 / E;
  static constexpr int RB = HD * E;
  static constexpr int CPR = RB / 16;
  static constexpr int KS = CPR / 2;
  static constexpr int CPT = 16 / CN;
};

template <typename T> __device__ __forceinline__ int u_swz(int row, int chunk) {
  if (BCfg<T>::CPR == 8) return chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
  return chunk ^ (row & 15);
}
template <typename T> __device__ __forceinline__ int tile_off(int row, int chunk) { return row * BCfg<T>::RB + (u_swz<T>(row, chunk) << 4); }

typedef short s16x4 __attribute__((ext_vector_type(4)));

// transposed A-operand chunk: element (k, i) = tile[row0 + krow(cc, g, k)][dt*32 + (lane & 31)], the k order being
// the accumulator-register order of the matching B operand (regs cc*CN .. cc*CN+CN-1).
template <typename T> __device__ __forceinline__ u32x4 load_t_chunk(const char* tile, int row0, int cc, int lane, int dt);
template <> __device__ __forceinline__ u32x4 load_t_chunk<float>(const char* tile, int row0, int cc, int lane, int dt) {
  const int d = dt * 32 + (lane & 31), r = row0 + 8 * cc + 4 * (lane >> 5);
  uint32_t v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = *(const uint32_t*)(tile + tile_off<float>(r + e, d >> 2) + ((d & 3) << 2));
  return mk4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ u32x2 tr_quad_u(const char* tile, int krow0, int lane, int dt) {
  const int p = lane & 15, seg = dt * 2 + ((lane >> 4) & 1);
  const int row = krow0 + (p >> 2);
  const int ch = seg * 2 + ((p >> 1) & 1);
  const char* a = tile + row * 128 + ((ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4) + ((p & 1) << 3);
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
  return __builtin_bit_cast(u32x2, r);
}
template <typename T> __device__ __forceinline__ u32x4 load_t_chunk16(const char* tile, int row0, int cc, int lane, int dt) {
  const int g = lane >> 5;
  const u32x2 a = tr_quad_u(tile, row0 + 16 * cc + 4 * g, lane, dt);
  const u32x2 b = tr_quad_u(tile, row0 + 16 * cc + 8 + 4 * g, lane, dt);
  const uint32_t ax = a.x, ay = a.y, bx = b.x, by = b.y;
  return mk4(ax, ay, bx, by);
}
template <> __device__ __forceinline__ u32x4 load_t_chunk<bf16_t>(const char* tile, int row0, int cc, int lane, int dt) { return load_t_chunk16<bf16_t>(tile, row0, cc, lane, dt); }
template <> __device__ __forceinline__ u32x4 load_t_chunk<f16_t>(const char* tile, int row0, int cc, int lane, int dt) { return load_t_chunk16<f16_t>(tile, row0, cc, lane, dt); }

template <typename T> __device__ __forceinline__ void store_quad_b(T* dst, const float* v) {
  if constexpr (sizeof(T) == 4) {
    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    u32x2 u;
    u.x = pack2(v[0], v[1], (T*)0);
    u.y = pack2(v[2], v[3], (T*)0);
    *(u32x2*)dst = u;
  }
}
// accumulator pair (2 d-tiles, C layout: column = token of this lane, rows = d) -> one token row of 64 values
template <typename T> __device__ __forceinline__ void store_row64(T* row, const f32x16 (&o)[2], int lane) {
  const int g = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float v[4] = {o[dt][4 * rq], o[dt][4 * rq + 1], o[dt][4 * rq + 2], o[dt][4 * rq + 3]};
      store_quad_b<T>(row + dt * 32 + 8 * rq + 4 * g, v);
    }
}

template <typename T>
__device__ __forceinline__ void stage_tile(char* tile, const T* src, int64_t ld, int rows_valid, int LP, int tid, int nthreads) {
  typedef BCfg<T> C;
  for (int c = tid; c < LP * C::CPR; c += nthreads) {
    const int row = c / C::CPR, ch = c - row * C::CPR;
    u32x4 v = mk4(0, 0, 0, 0);
    if (row < rows_valid) v = *(const u32x4*)(src + (int64_t)row * ld + ch * C::CN);
    *(u32x4*)(tile + tile_off<T>(row, ch)) = v;
  }
}

template <typename T, int NKT, int NW, bool GROUPED>
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, T* __restrict__ dqkv, int L, int H, float scale,
                                                           const float* __restrict__ key_bias, int Tn, int64_t total_rows, float drop_p,
                                                           uint32_t drop_seed) {
  typedef BCfg<T> C;
  constexpr int LP = NKT * 32;
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tA = smem;                 // K, then Q
  char* tB = smem + LP * C::RB;    // V, then dO
  float* Bs = (float*)(smem + 2 * LP * C::RB);
  float* Ls = Bs + LP;
  float* Ds = Ls + LP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int64_t row0 = (int64_t)b * L;
  const int Le = GROUPED ? (int)((total_rows - row0) < 32 ? (total_rows - row0) : 32) : L;
  const int64_t ldq = 3 * (int64_t)H * HD, ldo = (int64_t)H * HD;
  const T* qb = qkv + row0 * ldq + h * HD;
  const T* ob = out + row0 * ldo + h * HD;
  const T* dob = dout + row0 * ldo + h * HD;
  T* db = dqkv + row0 * ldq + h * HD;
  const float* lse_b = lse + ((int64_t)b * H + h) * L;
  const uint32_t dth = drop_thresh24(drop_p);
  const float dks = drop_seed ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t dbase = ((uint64_t)b * H + h) * (uint64_t)L;  // + q, then * L + key
  for (int c = tid; c < LP; c += NT) {
    Bs[c] = c < Le ? ((!GROUPED && key_bias) ? key_bias[(int64_t)b * L + c] : 0.f) : -INFINITY;
    Ls[c] = c < Le ? lse_b[c] : INFINITY;
  }
  stage_tile<T>(tA, qb + H * HD, ldq, Le, LP, tid, NT);
  stage_tile<T>(tB, qb + 2 * H * HD, ldq, Le, LP, tid, NT);
  __syncthreads();

  const int g = lane >> 5, ql = lane & 31;
  const int ntile = (Le + 31) >> 5;
  // ---------------------------------------------------------------- phase 1: dQ (lane = query)
  for (int qt = wave; qt < ntile; qt += NW) {
    const int q = qt * 32 + ql;
    const int qc = q < Le ? q : Le - 1;
    u32x4 qf[C::KS], dof[C::KS];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int off = (2 * ks + g) * C::CN;
      qf[ks] = *(const u32x4*)(qb + (int64_t)qc * ldq + off);
      dof[ks] = *(const u32x4*)(dob + (int64_t)qc * ldo + off);
      const u32x4 of = *(const u32x4*)(ob + (int64_t)qc * ldo + off);
      float a[C::CN], c2[C::CN];
      unpack_chunk<T>(dof[ks], a);
      unpack_chunk<T>(of, c2);
#pragma unroll
      for (int e = 0; e < C::CN; ++e) delta += a[e] * c2[e];
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lse_q = Ls[q];
    if (g == 0) Ds[q] = delta;
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < ntile) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        const int krow = kt * 32 + ql;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const u32x4 ka = *(const u32x4*)(tA + tile_off<T>(krow, 2 * ks + g));
          const u32x4 va = *(const u32x4*)(tB + tile_off<T>(krow, 2 * ks + g));
          mma_chunk<T>(s, ka, qf[ks]);
          mma_chunk<T>(dp, va, dof[ks]);
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 bq = *(const float4*)(Bs + kt * 32 + 8 * rq + 4 * g);
          const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            float p = expf(s[r] * scale + bb[e] - lse_q);
            if (GROUPED && ((8 * rq + 4 * g + e) / Tn != ql / Tn)) p = 0.f;
            float gd = dp[r];
            if (!GROUPED && drop_seed) gd = drop_keep(drop_seed, (dbase + qc) * L + kt * 32 + 8 * rq + 4 * g + e, dth) ? gd * dks : 0.f;
            s[r] = p * (gd - delta) * scale;  // dS^T
          }
        }
#pragma unroll
        for (int cc = 0; cc < C::CPT; ++cc) {
          float v[C::CN];
#pragma unroll
          for (int e = 0; e < C::CN; ++e) v[e] = s[cc * C::CN + e];
          const u32x4 bop = pack_chunk<T>(v);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) mma_chunk<T>(dq[dt], load_t_chunk<T>(tA, kt * 32, cc, lane, dt), bop);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (q < Le) store_row64<T>(db + (int64_t)q * ldq, dq, lane);
  }
  __syncthreads();
  // ---------------------------------------------------------------- phase 2: dK, dV (lane = key)
  stage_tile<T>(tA, qb, ldq, Le, LP, tid, NT);
  stage_tile<T>(tB, dob, ldo, Le, LP, tid, NT);
  __syncthreads();
  for (int kt = wave; kt < ntile; kt += NW) {
    const int key = kt * 32 + ql;
    const int kc = key < Le ? key : Le - 1;
    u32x4 kf[C::KS], vf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int off = (2 * ks + g) * C::CN;
      kf[ks] = *(const u32x4*)(qb + (int64_t)kc * ldq + H * HD + off);
      vf[ks] = *(const u32x4*)(qb + (int64_t)kc * ldq + 2 * H * HD + off);
    }
    const float kb = Bs[key];
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
#pragma unroll 1
    for (int qt = 0; qt < ntile; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
      const int qrow = qt * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const u32x4 qa = *(const u32x4*)(tA + tile_off<T>(qrow, 2 * ks + g));
        const u32x4 da = *(const u32x4*)(tB + tile_off<T>(qrow, 2 * ks + g));
        mma_chunk<T>(s, qa, kf[ks]);
        mma_chunk<T>(dp, da, vf[ks]);
      }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 lq = *(const float4*)(Ls + qt * 32 + 8 * rq + 4 * g);
        const float4 dq4 = *(const float4*)(Ds + qt * 32 + 8 * rq + 4 * g);
        const float ll[4] = {lq.x, lq.y, lq.z, lq.w}, dd[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rq + e;
          float p = expf(s[r] * scale + kb - ll[e]);
          if (GROUPED && ((8 * rq + 4 * g + e) / Tn != ql / Tn)) p = 0.f;
          float dm = 1.0f;
          if (!GROUPED && drop_seed) {
            const int qq = qt * 32 + 8 * rq + 4 * g + e;
            dm = drop_keep(drop_seed, (dbase + (qq < Le ? qq : Le - 1)) * L + key, dth) ? dks : 0.f;
          }
          s[r] = p * dm;                                // dropped P (feeds dV)
          dp[r] = p * (dm * dp[r] - dd[e]) * scale;     // dS
        }
      }
#pragma unroll
      for (int cc = 0; cc < C::CPT; ++cc) {
        float pv[C::CN], sv[C::CN];
#pragma unroll
        for (int e = 0; e < C::CN; ++e) {
          pv[e] = s[cc * C::CN + e];
          sv[e] = dp[cc * C::CN + e];
        }
        const u32x4 pb = pack_chunk<T>(pv), sb = pack_chunk<T>(sv);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          mma_chunk<T>(dv[dt], load_t_chunk<T>(tB, qt * 32, cc, lane, dt), pb);
          mma_chunk<T>(dk[dt], load_t_chunk<T>(tA, qt * 32, cc, lane, dt), sb);
        }
      }
    }
    if (key < Le) {
      store_row64<T>(db + (int64_t)key * ldq + H * HD, dk, lane);
      store_row64<T>(db + (int64_t)key * ldq + 2 * H * HD, dv, lane);
    }
  }
}

// ================================================================================================
// 16-bit full-attention backward, throughput form (same two phases and MFMA orientations as attn_bwd_kernel):
//  * tiles go global -> LDS by DMA with the swizzle applied on the source side; padded rows read a zero page;
//  * exp2 with log2(e) folded into the score scale; LDS holds -lse*log2(e) (or -inf for padded queries, which makes
//    their P rows exactly 0) and delta*scale, so a score costs FMA + v_exp_f32 and a dS costs FMA + MUL;
//  * the next query tile's Q / dO / O fragments are prefetched under the current tile's work (phase 1);
//  * dQ, dK, dV tiles are transposed through 4 KiB of wave-private LDS and leave as 16-byte row stores;
//  * <= 256 registers and ~76 KiB of LDS: two workgroups per CU.
__device__ u32x4 g_bwd_zero[4];
constexpr float LOG2E_B = 1.4426950408889634f;

// wave-private transpose: accumulator pair (column = token of this lane, rows = d) -> 32 row-major 128-byte rows.
// HALF: 2 KiB of staging instead of 4 -- the two 32-wide d halves go one after the other as 64-byte row pieces (used where
// the full staging would push the workgroup over half of the CU's LDS, i.e. 8 key tiles).
template <typename T, bool HALF = false>
__device__ __forceinline__ void store_rows_via_lds(char* Ow, const f32x16 (&o)[2], T* dst, int64_t ld, int row_base, int rows_valid, int lane) {
  const int g = lane >> 5, ql = lane & 31;
  if constexpr (!HALF) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint32_t lo = pack2(o[dt][4 * rq], o[dt][4 * rq + 1], (T*)0);
        const uint32_t hi = pack2(o[dt][4 * rq + 2], o[dt][4 * rq + 3], (T*)0);
        *(u32x2*)(Ow + ql * 128 + (((dt * 4 + rq) ^ ((ql >> 1) & 7)) << 4) + g * 8) = mk2(lo, hi);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = p * 8 + (lane >> 3), slot = lane & 7;
      const u32x4 v = *(const u32x4*)(Ow + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      if (row_base + row < rows_valid) __builtin_nontemporal_store(v, (u32x4*)(dst + (int64_t)(row_base + row) * ld + slot * 8));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging rows may be rewritten right away
  } else {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint32_t lo = pack2(o[dt][4 * rq], o[dt][4 * rq + 1], (T*)0);
        const uint32_t hi = pack2(o[dt][4 * rq + 2], o[dt][4 * rq + 3], (T*)0);
        *(u32x2*)(Ow + ql * 64 + ((rq ^ ((ql >> 2) & 3)) << 4) + g * 8) = mk2(lo, hi);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = p * 16 + (lane >> 2), slot = lane & 3;
        const u32x4 v = *(const u32x4*)(Ow + row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
        if (row_base + row < rows_valid) __builtin_nontemporal_store(v, (u32x4*)(dst + (int64_t)(row_base + row) * ld + dt * 32 + slot * 8));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

template <typename T, int NKT, bool HAS_BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd16_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                          const T* __restrict__ dout, const float* __restrict__ lse,
                                                                          T* __restrict__ dqkv, int L, int H, float scale,
                                                                          const float* __restrict__ key_bias, float drop_p, uint32_t drop_seed, int order) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int LP = NKT * 32, RB = 128;
  constexpr bool HALF = NKT == 8;             // 8 key tiles: 2 KiB staging per wave keeps two workgroups per CU (2 x 75 KiB)
  constexpr int OW = HALF ? 2048 : 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tA = smem;              // K, then Q
  char* tB = smem + LP * RB;    // V, then dO
  char* Os = smem + 2 * LP * RB;
  float* Bs = (float*)(Os + 4 * OW);  // key bias * log2(e); -inf on padded keys
  float* Ls = Bs + LP;                   // -lse * log2(e); -inf on padded queries
  float* Ds = Ls + LP;                   // delta * scale
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, h;
  attn_unit(blockIdx.x, gridDim.x, H, order, b, h);
  const int64_t row0 = (int64_t)b * L;
  const int64_t ldq = 3 * (int64_t)H * HD, ldo = (int64_t)H * HD;
  const T* qb = qkv + row0 * ldq + h * HD;
  const T* ob = out + row0 * ldo + h * HD;
  const T* dob = dout + row0 * ldo + h * HD;
  T* db = dqkv + row0 * ldq + h * HD;
  const float* lse_b = lse + ((int64_t)b * H + h) * L;
  const uint32_t dth = drop_thresh24(drop_p);
  const float dks = drop_seed ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t dbase = ((uint64_t)b * H + h) * (uint64_t)L;  // + q, then * L + key
  const float sl = scale * LOG2E_B;
  for (int c = tid; c < LP; c += 256) {
    Bs[c] = c < L ? (HAS_BIAS ? key_bias[(int64_t)b * L + c] * LOG2E_B : 0.f) : -INFINITY;
    Ls[c] = c < L ? -lse_b[c] * LOG2E_B : -INFINITY;
  }
  const uint32_t a_lds = lds_addr_of(tA), b_lds = lds_addr_of(tB);
  const char* zero = (const char*)g_bwd_zero;
  // two row-major (row, 64) tiles -> LDS images, chunk ^ (bit1(row) << 2 | (row >> 2) & 3)
  auto stage2 = [&](const T* srcA, int64_t lda_, const T* srcB, int64_t ldb_) {
#pragma unroll
    for (int i = 0; i < NKT; ++i) {
      const int piece = wave + 4 * i;
      const int row = piece * 8 + (lane >> 3), slot = lane & 7;
      const int ch = slot ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
      const char* pa = row < L ? (const char*)(srcA + (int64_t)row * lda_ + ch * 8) : zero;
      const char* pb = row < L ? (const char*)(srcB + (int64_t)row * ldb_ + ch * 8) : zero;
      dma16(pa, __builtin_amdgcn_readfirstlane(a_lds + piece * 1024));
      dma16(pb, __builtin_amdgcn_readfirstlane(b_lds + piece * 1024));
    }
  };
  stage2(qb + H * HD, ldq, qb + 2 * H * HD, ldq);

  const int g = lane >> 5, ql = lane & 31;
  const int ntile = (L + 31) >> 5;
  char* Ow = Os + wave * OW;
  // ---------------------------------------------------------------- phase 1: dQ (lane = query)
  auto load_q3 = [&](int qt, u32x4(&qf)[4], u32x4(&dof)[4], u32x4(&of)[4]) {
    const int qc = min(qt * 32 + ql, L - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = (2 * ks + g) * 8;
      qf[ks] = *(const u32x4*)(qb + (int64_t)qc * ldq + off);
      dof[ks] = *(const u32x4*)(dob + (int64_t)qc * ldo + off);
      of[ks] = *(const u32x4*)(ob + (int64_t)qc * ldo + off);
    }
  };
  u32x4 qf[4], dof[4], of[4];
  load_q3(min(wave, ntile - 1), qf, dof, of);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int qt = wave; qt < ntile; qt += 4) {
    const int q = qt * 32 + ql;
    const int qc = min(q, L - 1);
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[8], c2[8];
      unpack_chunk<T>(dof[ks], a);
      unpack_chunk<T>(of[ks], c2);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += a[e] * c2[e];
    }
    delta += __shfl_xor(delta, 32, 64);
    const float nds = -delta * scale;
    const float nlq = Ls[q];
    if (g == 0) Ds[q] = nds;
    u32x4 qn[4], don[4], on[4];
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < ntile) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        const int krow = kt * 32 + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 ka = *(const u32x4*)(tA + tile_off<T>(krow, 2 * ks + g));
          const u32x4 va = *(const u32x4*)(tB + tile_off<T>(krow, 2 * ks + g));
          mma_chunk<T>(s, ka, qf[ks]);
          mma_chunk<T>(dp, va, dof[ks]);
        }
        if (kt == 0) load_q3(min(qt + 4, ntile - 1), qn, don, on);  // lands under this tile's work
        const bool plain = !HAS_BIAS && (kt + 1) * 32 <= L;          // all 32 keys valid, no bias
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float bb[4] = {nlq, nlq, nlq, nlq};
          if (!plain) {
            const float4 bq = *(const float4*)(Bs + kt * 32 + 8 * rq + 4 * g);
            bb[0] += bq.x; bb[1] += bq.y; bb[2] += bq.z; bb[3] += bq.w;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sl, bb[e]));
            float gd = dp[r];
            if constexpr (DROP) gd = drop_keep(drop_seed, (dbase + qc) * L + kt * 32 + 8 * rq + 4 * g + e, dth) ? gd * dks : 0.f;
            s[r] = p * fmaf(gd, scale, nds);  // dS^T = P o (dP - delta) * scale
          }
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = s[cc * 8 + e];
          const u32x4 bop = pack_chunk<T>(v);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) mma_chunk<T>(dq[dt], load_t_chunk<T>(tA, kt * 32, cc, lane, dt), bop);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    store_rows_via_lds<T, HALF>(Ow, dq, db, ldq, qt * 32, L, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = qn[ks];
      dof[ks] = don[ks];
      of[ks] = on[ks];
    }
  }
  __syncthreads();
  // ---------------------------------------------------------------- phase 2: dK, dV (lane = key)
  stage2(qb, ldq, dob, ldo);
  u32x4 kf[4], vf[4];
  {
    const int kc = min(min(wave, ntile - 1) * 32 + ql, L - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = (2 * ks + g) * 8;
      kf[ks] = *(const u32x4*)(qb + (int64_t)kc * ldq + H * HD + off);
      vf[ks] = *(const u32x4*)(qb + (int64_t)kc * ldq + 2 * H * HD + off);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = wave; kt < ntile; kt += 4) {
    const int key = kt * 32 + ql;
    if (kt != wave) {
      const int kc = min(key, L - 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = (2 * ks + g) * 8;
        kf[ks] = *(const u32x4*)(qb + (int64_t)kc * ldq + H * HD + off);
        vf[ks] = *(const u32x4*)(qb + (int64_t)kc * ldq + 2 * H * HD + off);
      }
    }
    const float kb = HAS_BIAS ? Bs[key] : 0.f;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
#pragma unroll 1
    for (int qt = 0; qt < ntile; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
      const int qrow = qt * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 qa = *(const u32x4*)(tA + tile_off<T>(qrow, 2 * ks + g));
        const u32x4 da = *(const u32x4*)(tB + tile_off<T>(qrow, 2 * ks + g));
        mma_chunk<T>(s, qa, kf[ks]);
        mma_chunk<T>(dp, da, vf[ks]);
      }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 lq = *(const float4*)(Ls + qt * 32 + 8 * rq + 4 * g);
        const float4 dq4 = *(const float4*)(Ds + qt * 32 + 8 * rq + 4 * g);
        float ll[4] = {lq.x, lq.y, lq.z, lq.w};
        const float dd[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
        if (HAS_BIAS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ll[e] += kb;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rq + e;
          const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sl, ll[e]));
          float gd = dp[r], pm = p;
          if constexpr (DROP) {
            const int qq = qt * 32 + 8 * rq + 4 * g + e;
            const bool keep = drop_keep(drop_seed, (dbase + (qq < L ? qq : L - 1)) * L + key, dth);
            pm = keep ? p * dks : 0.f;
            gd = keep ? gd * dks : 0.f;
          }
          s[r] = pm;                          // dropped P (feeds dV)
          dp[r] = p * fmaf(gd, scale, dd[e]);  // dS
        }
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        float pv[8], sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = s[cc * 8 + e];
          sv[e] = dp[cc * 8 + e];
        }
        const u32x4 pb = pack_chunk<T>(pv), sb = pack_chunk<T>(sv);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          mma_chunk<T>(dv[dt], load_t_chunk<T>(tB, qt * 32, cc, lane, dt), pb);
          mma_chunk<T>(dk[dt], load_t_chunk<T>(tA, qt * 32, cc, lane, dt), sb);
        }
      }
    }
    store_rows_via_lds<T, HALF>(Ow, dk, db + H * HD, ldq, kt * 32, L, lane);
    store_rows_via_lds<T, HALF>(Ow, dv, db + 2 * H * HD, ldq, kt * 32, L, lane);
  }
}

#ifdef ALPRO_ABLATIONS
// measurement build only: shader-clock stamps of one workgroup (alpro_debug_attn_bwd_stamps), 16 per wave
__device__ unsigned long long g_attn_ts[8 * 16];
__device__ int g_attn_ts_block = -1;
__device__ int g_attn_alias = 0;   // > 0: every unit reads / writes unit (u mod alias) -- wrong results, all inputs L2 resident (timing experiment)
#define ALPRO_TS(i)                                                                                           \
  do {                                                                                                        \
    if ((int)blockIdx.x == g_attn_ts_block && lane == 0) g_attn_ts[wave * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define ALPRO_TS(i) do { } while (0)
#endif

// ================================================================================================
// Key-owned 16-bit backward for 5..8 key tiles (ViT spatial L = 197, fusion L = 237): every (query tile, key tile)
// pair is computed ONCE.  The two-phase kernel above evaluates S, dP, P and dS of every pair twice (once per
// orientation, 28 MFMAs + 2 x 16 exp per pair); here wave w owns key tile w for the whole unit (K / V fragments and
// the dK^T / dV^T accumulators stay in its registers) and walks the query tiles in lockstep with the other waves:
//       S = Q K^T, dP = dO V^T, P, dS                       as phase 2 above (lane = key)
//       dV^T += dO^T P,  dK^T += Q^T dS                     as phase 2 above
//       dQ^T(partial over this key tile) = K^T dS^T         A = K^T via transpose reads (as phase 1), B = dS^T:
//           the dS tile is written 16-bit into 2 KiB of wave-private LDS as [key][query] rows (the same packed pairs
//           that feed the dK product) and read back with ds_read_b64_tr_b16 in the k order of the A chunk
// 20 MFMAs and 16 exp per pair, and the dQ contraction over ALL keys happens in one accumulator, so nothing is reduced across
// waves: the workgroup alternates between
//   main pass  (RQ query tiles): wave w computes the pairs (q tile, key tile w), accumulates dV^T / dK^T in registers and writes
//              each dS tile, 16-bit, as a [key][query] image of 2 KiB into shared LDS (the packed pairs that feed the dK product,
//              8-byte slots swizzled by (key >> 2) & 7: conflict-free for the ds_write_b64 rows and for the transpose reads);
//   dQ pass    wave j takes (query tile j >> 1, d half j & 1): dQ^T = sum over key tiles K^T dS^T with A = K^T via
//              ds_read_b64_tr_b16 (as phase 1 above) and B = dS^T read back from the images with ds_read_b64_tr_b16 in the k
//              order of the A chunk; two alternating accumulators, fixed order (deterministic, no atomics); rows leave as 8-byte
//              stores straight from the accumulator layout.
// Two workgroup barriers per round (RQ = 4 query tiles with <= 7 key tiles, 3 with 8).  V is never staged (only a register
// operand); K, Q, dO tiles + the dS images take 143..147 KiB: one 8-wave workgroup per CU.  Dropout is a template parameter here
// (and in the two-phase kernel): a runtime test per score splits the softmax into 16 basic blocks per pair and serialises the
// v_exp_f32 latencies.
template <typename T, int NKT, bool DROP>
__global__ __launch_bounds__(512) void attn_bwd16k_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                        const T* __restrict__ dout, const float* __restrict__ lse,
                                                                        T* __restrict__ dqkv, int L, int H, float scale,
                                                                        const float* __restrict__ key_bias, float drop_p, uint32_t drop_seed, int order) {
  static_assert(sizeof(T) == 2 && NKT >= 5 && NKT <= 8, "16-bit storage, 5..8 key tiles");
  constexpr int LP = NKT * 32, RB = 128;
  constexpr int RQ = NKT <= 7 ? 4 : 3;   // query tiles per round
  constexpr int IMG = 2048;               // one dS tile: 32 key rows x 32 queries x 2 bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tK = smem;
  char* tQ = smem + LP * RB;
  char* tD = smem + 2 * LP * RB;       // dO
  char* DSb = smem + 3 * LP * RB;      // dS images [query tile of the round][key tile]; at the end the row staging of dK / dV
  float* Bs = (float*)(DSb + RQ * NKT * IMG);  // key bias * log2(e); -inf on padded keys
  float* Ls = Bs + LP;                 // -lse * log2(e); -inf on padded queries
  float* Ds = Ls + LP;                 // -delta * scale
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, h;
  attn_unit(blockIdx.x, gridDim.x, H, order, b, h);
  const int64_t row0 = (int64_t)b * L;
  const int64_t ldq = 3 * (int64_t)H * HD, ldo = (int64_t)H * HD;
  const T* qb = qkv + row0 * ldq + h * HD;
  const T* ob = out + row0 * ldo + h * HD;
  const T* dob = dout + row0 * ldo + h * HD;
  T* db = dqkv + row0 * ldq + h * HD;
  const float* lse_b = lse + ((int64_t)b * H + h) * L;
  const uint32_t dth = drop_thresh24(drop_p);
  const float dks = drop_seed ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t dbase = ((uint64_t)b * H + h) * (uint64_t)L;  // + q, then * L + key
  const float sl = scale * LOG2E_B;
  const int g = lane >> 5, ql = lane & 31;
  const int ntile = (L + 31) >> 5;           // >= 5 (dispatch)
  const bool active = wave < ntile;          // wave-uniform: this wave owns key tile `wave`
  ALPRO_TS(0);
  for (int c = tid; c < LP; c += 512) {
    Bs[c] = c < L ? (key_bias ? key_bias[(int64_t)b * L + c] * LOG2E_B : 0.f) : -INFINITY;
    Ls[c] = c < L ? -lse_b[c] * LOG2E_B : -INFINITY;
  }
  {  // K, Q, dO rows -> LDS images (chunk ^ (bit1(row) << 2 | (row >> 2) & 3)), 1 KiB DMA pieces
    const uint32_t k_lds = lds_addr_of(tK), q_lds = lds_addr_of(tQ), d_lds = lds_addr_of(tD);
    const char* zero = (const char*)g_bwd_zero;
#pragma unroll
    for (int i = 0; i < (NKT * 4 + 7) / 8; ++i) {
      const int piece = wave + 8 * i;
      if (piece < ntile * 4) {
        const int row = piece * 8 + (lane >> 3), slot = lane & 7;
        const int ch = slot ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
        const char* pk = row < L ? (const char*)(qb + (int64_t)row * ldq + H * HD + ch * 8) : zero;
        const char* pq = row < L ? (const char*)(qb + (int64_t)row * ldq + ch * 8) : zero;
        const char* pd = row < L ? (const char*)(dob + (int64_t)row * ldo + ch * 8) : zero;
        dma16(pk, __builtin_amdgcn_readfirstlane(k_lds + piece * 1024));
        dma16(pq, __builtin_amdgcn_readfirstlane(q_lds + piece * 1024));
        dma16(pd, __builtin_amdgcn_readfirstlane(d_lds + piece * 1024));
      }
    }
  }
  // K / V fragments of key tile `wave` (B operands, lane = key) and delta of QUERY tile `wave`
  u32x4 kf[4], vf[4];
  {
    const int rc = min(min(wave, ntile - 1) * 32 + ql, L - 1);
    u32x4 dof[4], of[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = (2 * ks + g) * 8;
      dof[ks] = *(const u32x4*)(dob + (int64_t)rc * ldo + off);
      of[ks] = *(const u32x4*)(ob + (int64_t)rc * ldo + off);
      kf[ks] = *(const u32x4*)(qb + (int64_t)rc * ldq + H * HD + off);
      vf[ks] = *(const u32x4*)(qb + (int64_t)rc * ldq + 2 * H * HD + off);
    }
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[8], c2[8];
      unpack_chunk<T>(dof[ks], a);
      unpack_chunk<T>(of[ks], c2);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += a[e] * c2[e];
    }
    delta += __shfl_xor(delta, 32, 64);
    if (active && g == 0) Ds[wave * 32 + ql] = -delta * scale;
  }
  ALPRO_TS(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ALPRO_TS(2);
  __syncthreads();
  ALPRO_TS(3);

  const int key = wave * 32 + ql;
  const bool need_kb = key_bias != nullptr || (wave + 1) * 32 > L;   // padded keys of the last tile: P = exp2(-inf) = 0, so they add nothing to dQ
  const float kb = (active && need_kb) ? Bs[key] : 0.f;
  const int swk = (ql >> 2) & 7;                           // 8-byte slot swizzle of the dS image (row = key)
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
  // dS[query ql of the tile][keys k0 .. k0+3] from a [key][query] image (k0 multiple of 4)
  auto ds_quad = [&](const char* img, int k0) -> u32x2 {
    const int p = lane & 15, cb = (lane >> 4) & 1;
    const int row = k0 + (p >> 2);
    const char* a = img + row * 64 + ((((cb << 2) | (p & 3)) ^ ((row >> 2) & 7)) << 3);
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    return __builtin_bit_cast(u32x2, r);
  };
#pragma unroll 1
  for (int q0 = 0; q0 < ntile; q0 += RQ) {
    const int nq = min(RQ, ntile - q0);
    // ---------------------------------------------------------------- main pass: pairs (q0 + tl, key tile `wave`)
    if (active) {
#pragma unroll 1
      for (int tl = 0; tl < nq; ++tl) {
        const int qt = q0 + tl;
        char* img = DSb + (tl * NKT + wave) * IMG;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        const int qrow = qt * 32 + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 qa = *(const u32x4*)(tQ + tile_off<T>(qrow, 2 * ks + g));
          const u32x4 da = *(const u32x4*)(tD + tile_off<T>(qrow, 2 * ks + g));
          mma_chunk<T>(s, qa, kf[ks]);
          mma_chunk<T>(dp, da, vf[ks]);
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 lq = *(const float4*)(Ls + qt * 32 + 8 * rq + 4 * g);
          const float4 dq4 = *(const float4*)(Ds + qt * 32 + 8 * rq + 4 * g);
          float ll[4] = {lq.x, lq.y, lq.z, lq.w};
          const float dd[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
          if (need_kb) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ll[e] += kb;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sl, ll[e]));
            float gd = dp[r], pm = p;
            if constexpr (DROP) {
              const int qq = qt * 32 + 8 * rq + 4 * g + e;
              const bool keep = drop_keep(drop_seed, (dbase + (qq < L ? qq : L - 1)) * L + (key < L ? key : L - 1), dth);
              pm = keep ? p * dks : 0.f;
              gd = keep ? gd * dks : 0.f;
            }
            s[r] = pm;                           // dropped P (feeds dV)
            dp[r] = p * fmaf(gd, scale, dd[e]);  // dS
          }
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          float pv[8], sv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pv[e] = s[cc * 8 + e];
            sv[e] = dp[cc * 8 + e];
          }
          const u32x4 pb = pack_chunk<T>(pv), sb = pack_chunk<T>(sv);
          // dS image: row = key (64 bytes = 32 queries), registers 4rq .. 4rq+3 are queries 8rq + 4g .. +3 -> 8-byte slot 2rq + g
          *(u32x2*)(img + ql * 64 + (((4 * cc + g) ^ swk) << 3)) = mk2(sb.x, sb.y);
          *(u32x2*)(img + ql * 64 + (((4 * cc + 2 + g) ^ swk) << 3)) = mk2(sb.z, sb.w);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            mma_chunk<T>(dv[dt], load_t_chunk<T>(tD, qt * 32, cc, lane, dt), pb);
            mma_chunk<T>(dk[dt], load_t_chunk<T>(tQ, qt * 32, cc, lane, dt), sb);
          }
        }
      }
    }
    ALPRO_TS(4 + (q0 ? 4 : 0));
    __syncthreads();
    ALPRO_TS(5 + (q0 ? 4 : 0));
    // ---------------------------------------------------------------- dQ pass: wave j = (query tile j >> 1, d half j & 1)
    if (wave < 2 * nq) {
      const int tl = wave >> 1, dt = wave & 1;
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        if (kt < ntile) {
          const char* img = DSb + (tl * NKT + kt) * IMG;
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const u32x2 lo = ds_quad(img, 16 * cc + 4 * g), hi = ds_quad(img, 16 * cc + 8 + 4 * g);
            const uint32_t lx = lo.x, ly = lo.y, hx = hi.x, hy = hi.y;
            mma_chunk<T>(acc[kt & 1], load_t_chunk<T>(tK, kt * 32, cc, lane, dt), mk4(lx, ly, hx, hy));
          }
        }
      const int q = (q0 + tl) * 32 + ql;
      if (q < L) {
        T* dst = db + (int64_t)q * ldq + dt * 32 + 4 * g;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const uint32_t lo = pack2(acc[0][4 * rq] + acc[1][4 * rq], acc[0][4 * rq + 1] + acc[1][4 * rq + 1], (T*)0);
          const uint32_t hi = pack2(acc[0][4 * rq + 2] + acc[1][4 * rq + 2], acc[0][4 * rq + 3] + acc[1][4 * rq + 3], (T*)0);
          __builtin_nontemporal_store(mk2(lo, hi), (u32x2*)(dst + 8 * rq));
        }
      }
    }
    ALPRO_TS(6 + (q0 ? 4 : 0));
    __syncthreads();   // the images are rewritten by the next round / become the row staging below
    ALPRO_TS(7 + (q0 ? 4 : 0));
  }
  if (active) {
    store_rows_via_lds<T, false>(DSb + wave * 4096, dk, db + H * HD, ldq, wave * 32, L, lane);
    store_rows_via_lds<T, false>(DSb + wave * 4096, dv, db + 2 * H * HD, ldq, wave * 32, L, lane);
  }
  ALPRO_TS(12);
}

#ifdef ALPRO_ABLATIONS   // measurement build only (round 4: slower than the shipped kernels on every shape measured, profiles/r3_attn_bwd_phase_stamps.txt)
// ================================================================================================
// Persistent variant of the key-owned backward for EXACTLY 7 key tiles, no key bias, no dropout (ViT spatial attention,
// L = 197): one 8-wave workgroup per CU walks its units (sequence, head) and the NEXT query tiles are always in flight.
// The non-persistent kernel above spends a third of every unit waiting for its 126 KiB of inputs (phase stamps in
// profiles/r3_attn_bwd_phase_stamps.txt: ~16k of ~45k cycles) with nothing to compute -- one workgroup per CU, so no
// neighbour hides it.  Here:
//   * Q / dO / O rows travel as 12 KiB query-tile records through a ring of NS slots, DMAed NS-1 steps ahead of use; K of
//     the next unit goes into the other half of a K double buffer during steps 1..4 of the current unit;
//   * waves 0..6 own one key tile each (as above) and do one pair per step; V / K fragments of the next unit are reloaded
//     into kf / vf right after their last use in the unit's last pair;
//   * wave 7 (idle above) is the service wave: the whole dQ tile of the PREVIOUS step (28 MFMAs over the seven dS images
//     of that step; images are double buffered by step parity) and the row statistics of the NEXT step's query tile
//     (-lse*log2e and -delta*scale from the dO / O rows of its ring record);
//   * one workgroup barrier per step.  DMA completion is tracked per wave with s_waitcnt vmcnt(n), n = this wave's own
//     copies issued after the ones that must have landed (every wave issues a fixed share per step, so n is a function of
//     the step position only); compiler-visible loads / stores in between only make that wait stricter, never laxer.
// Deterministic (fixed summation orders, no atomics); dK / dV / dQ leave as 8-byte stores straight from the accumulators.
template <typename T>
__global__ __launch_bounds__(512) void attn_bwd16p_kernel(const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
                                                          const float* __restrict__ lse, T* __restrict__ dqkv, int L, int H, float scale,
                                                          int units, int flags) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int NT = 7, NS = 5, KB = NT * 4096, IMG = 2048, REC = 3 * 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kb = smem;                       // 2 x K tile images
  char* Im = Kb + 2 * KB;                // 2 (step parity) x 7 dS images
  char* Rg = Im + 2 * NT * IMG;          // NS records: Q | dO | O rows of one query tile
  float* St = (float*)(Rg + NS * REC);   // 2 (step parity) x {Ls[32], Ds[32]}
  float* Lq = St + 2 * 64;               // 8 x lse[32] of the coming query tiles (service wave's own queue; registers would be live in the pair path too)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, ql = lane & 31;
  const int nu = (units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // units of this workgroup: blockIdx + i * grid
  const int S = nu * NT;                                                           // steps
  const int64_t ldq = 3 * (int64_t)H * HD, ldo = (int64_t)H * HD;
  const float sl = scale * LOG2E_B;
  const uint32_t kb_lds = lds_addr_of(Kb), rg_lds = lds_addr_of(Rg);
  // Per-lane LDS offsets: four base values; every access is (base ^ constant) + wave-uniform base + constant.  The tile swizzles only
  // touch row bits 1..3 and XOR disjoint bit fields, so the per-chunk / per-row-block variants are single v_xor rematerialisations of the
  // bases instead of ~20 live address registers (which spilled: a spill reload is a vector-memory load, and loads return in order --
  // it would wait for every DMA prefetch issued before it).
  const int p16 = lane & 15, r0 = 4 * g + (p16 >> 2);
  const int fo0 = ql * 128 + ((g ^ ((((ql >> 1) & 1) << 2) | ((ql >> 2) & 3))) << 4);          // fragment chunk 2ks + g of row ql: fo0 ^ (ks << 5)
  const int tro0 = r0 * 128 + (((2 * ((lane >> 4) & 1) + ((p16 >> 1) & 1)) ^ ((((r0 >> 1) & 1) << 2) | (r0 >> 2))) << 4) + ((p16 & 1) << 3);
  const int iw0 = ql * 64 + ((g ^ ((ql >> 2) & 7)) << 3);                                     // dS image write, slot 4cc + 2h + g: iw0 ^ ((4cc + 2h) << 3)
  const int ir0 = r0 * 64 + ((((((lane >> 4) & 1) << 2) | (p16 & 3)) ^ (r0 >> 2)) << 3);      // dS image transpose read, rows 16cc + 8h + r0
  // copy source, lane part: a 1 KiB piece is 8 rows x 8 chunks; lane l copies chunk (l & 7) ^ swizzle(row) of row (l >> 3) of the piece.
  // With row = 8 * piece + (l >> 3): swizzle = ((l >> 4) & 1) << 2 | ((2 * piece + (l >> 5)) & 3) = c0 ^ (2 * (piece & 1))
  const int dm0 = ((lane & 7) ^ ((((lane >> 4) & 1) << 2) | (lane >> 5))) << 4;              // byte offset in the 128-byte row: dm0 ^ ((piece & 1) << 5)
  // The bases pass through an empty asm at the top of every step (`fresh`): without it the compiler hoists all ~20 variants out of the
  // step loop as invariants and then spills them, instead of recomputing one v_xor at the use.
  int fb = fo0, tb = tro0, wb = iw0, rb = ir0, db = dm0, lz = lane;
  auto fresh = [&] { asm volatile("" : "+v"(fb), "+v"(tb), "+v"(wb), "+v"(rb), "+v"(db), "+v"(lz)); };
  auto fo = [&](int ks) { return fb ^ (ks << 5); };
  auto tro = [&](int h2, int dt) { return (tb ^ (((4 * dt) ^ (2 * h2)) << 4)) + h2 * 1024; };   // rows 8h + r0, 16-column segment 2dt + (lane >> 4 & 1)
  auto iw = [&](int cc, int h2) { return wb ^ ((4 * cc + 2 * h2) << 3); };
  auto ir = [&](int cc, int h2) { return (rb ^ ((4 * cc + 2 * h2) << 3)) + (16 * cc + 8 * h2) * 64; };
  auto tr8 = [](const char* a) -> u32x2 {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    return __builtin_bit_cast(u32x2, r);
  };
  // transposed A chunk of rows [rowbase, rowbase + 16) (rowbase a multiple of 16), d half dt (the chunk the other kernels' transposed-chunk loader returns)
  auto tr_chunk = [&](const char* tile, int rowbase, int dt) -> u32x4 {
    const u32x2 x = tr8(tile + tro(0, dt) + rowbase * 128), y = tr8(tile + tro(1, dt) + rowbase * 128);
    const uint32_t x0 = x.x, x1 = x.y, y0 = y.x, y1 = y.y;
    return mk4(x0, x1, y0, y1);
  };
  // (sequence b, head h) of four units at a time (previous, current, next, the one after), renewed once per unit; base pointers are a
  // few scalar multiply-adds from them at the use.  Computing u / H and five 64-bit products at every use was most of the ~1700 cycles a
  // wave spent "issuing" two copies (profiles/r3_attn_bwd_phase_stamps.txt); whole pointer sets for four units do not fit the scalar
  // registers.  Beyond the last unit they repeat it (harmless re-reads keep the copy counts fixed).
  struct Unit { uint32_t b, h; };
  const uint32_t h_magic = (uint32_t)(((1ull << 32) + (uint32_t)H - 1) / (uint32_t)H);   // u / H == umulhi(u, h_magic) while u * H < 2^32
  auto unit_at = [&](int ui) -> Unit {
    uint32_t u = blockIdx.x + (uint32_t)(ui < nu ? ui : nu - 1) * gridDim.x;
#ifdef ALPRO_ABLATIONS
    if (g_attn_alias > 0) u %= (uint32_t)g_attn_alias;
#endif
    const uint32_t ub = __umulhi(u, h_magic);
    return Unit{ub, u - ub * (uint32_t)H};
  };
  const int64_t seq_o = (int64_t)L * ldo;   // elements of one sequence in out / dout; qkv / dqkv: 3 x
  auto pq = [&](const Unit& U) { return qkv + ((int64_t)U.b * seq_o * 3 + U.h * HD); };
  auto pg = [&](const Unit& U) { return dqkv + ((int64_t)U.b * seq_o * 3 + U.h * HD); };
  auto po = [&](const Unit& U) { return out + ((int64_t)U.b * seq_o + U.h * HD); };
  auto pd = [&](const Unit& U) { return dout + ((int64_t)U.b * seq_o + U.h * HD); };
  auto pl = [&](const Unit& U) { return lse + ((int64_t)U.b * H + U.h) * L; };
  Unit Up = unit_at(0), Uc = Up, Un = unit_at(1), Unn = unit_at(2);
  // ---- copies.  Source = wave-uniform base + 32-bit lane offset; padded rows (>= L) repeat row L-1: whatever they hold is multiplied
  // by P = 0 / dS = 0 (padded queries have Ls = -inf, padded keys start S at -inf) or never stored.
  auto copy_piece = [&](const T* base, int64_t ld, int row0, int piece, uint32_t dst) {
    const int row = min(row0 + piece * 8 + (lz >> 3), L - 1);
    const uint32_t off = (uint32_t)row * (uint32_t)(ld * 2) + (uint32_t)(db ^ ((piece & 1) << 5));
    const uint64_t bp = (uint64_t)base;
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)bp);
    // reserved-register site (the product is built with -Werror=inline-asm; this one is deliberate): global_load_lds takes its LDS address from m0; listing it as clobbered is what keeps the compiler from assuming a value of its own survives the statement (it writes m0 itself before each of its own uses: LDS-DMA builtins, s_movrel); the K-loop ISA tests of tests/test_host_cpu.py read the built object
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(sb), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
  };
  // record of query tile tx of unit U -> ring slot: 12 pieces, this wave's are o = wave, wave + 8 -> array o >> 2 (Q, dO, O), piece o & 3
  auto issue_record_op = [&](const Unit& U, int tx, int slot, int i) {   // i = 0, 1: this wave's first / second piece
    const uint32_t dst = rg_lds + (uint32_t)slot * REC;
    const int o = wave + 8 * i;
    if (o < 12) {
      const int arr = o >> 2, p = o & 3;
      copy_piece(arr == 0 ? pq(U) : (arr == 1 ? pd(U) : po(U)), arr == 0 ? ldq : ldo, tx * 32, p, __builtin_amdgcn_readfirstlane(dst + arr * 4096 + p * 1024));
    }
  };
  auto issue_record = [&](const Unit& U, int tx, int slot) {
    issue_record_op(U, tx, slot, 0);
    issue_record_op(U, tx, slot, 1);
  };
  const int rec_ops = wave < 4 ? 2 : 1;
  // K of unit U -> Kb[par]; 28 pieces, 7 per call (part 0..3), one per wave 0..6
  auto issue_k = [&](const Unit& U, int par, int part) {
    if (wave < 7) {
      const int piece = part * 7 + wave;
      copy_piece(pq(U) + H * HD, ldq, 0, piece, __builtin_amdgcn_readfirstlane(kb_lds + (uint32_t)par * KB + piece * 1024));
    }
  };
  const int k_ops = wave < 7 ? 1 : 0;
  auto ops_at = [&](int t) { return rec_ops + ((t >= 1 && t <= 4) ? k_ops : 0); };   // copies this wave issues in a step at position t
  auto wait_vm = [&](int n) {   // n is wave-uniform, 0..6
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
  };
  auto block_sync = [] {   // barrier that does not drain vmcnt
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // row statistics of the query tile in ring slot `slot` (tile tx of its unit) -> St[par]; service wave only
  auto stats = [&](int slot, int tx, int par, float lse_v) {
    const char* rec = Rg + slot * REC;
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[8], c2[8];
      unpack_chunk<T>(*(const u32x4*)(rec + 4096 + fo(ks)), a);
      unpack_chunk<T>(*(const u32x4*)(rec + 8192 + fo(ks)), c2);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += a[e] * c2[e];
    }
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0) {
      float* st = St + par * 64;
      st[ql] = tx * 32 + ql < L ? -lse_v * LOG2E_B : -INFINITY;
      st[32 + ql] = -delta * scale;
    }
  };

  // ---------------------------------------------------------------- fill: K of unit 0, records 0 .. NS-2 (all of unit 0: NS-1 <= NT)
#pragma unroll
  for (int part = 0; part < 4; ++part) issue_k(Uc, 0, part);
#pragma unroll
  for (int x = 0; x < NS - 1; ++x) issue_record(Uc, x, x);
  u32x4 kf[4], vf[4];
  const int key = wave * 32 + ql;          // pair waves: the key of this lane
  const float s_init = key < L ? 0.f : -INFINITY;   // padded keys: S = -inf -> P = 0 -> no contribution to dQ
  if (wave == 7) {
    float l[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) l[x] = pl(Uc)[min(x * 32 + ql, L - 1)];
    if (g == 0) {
#pragma unroll
      for (int x = 1; x < 4; ++x) Lq[x * 32 + ql] = l[x];
    }
    wait_vm(2 * rec_ops);   // K, records 0 and 1 (records 2, 3 may still fly)
    block_sync();
    stats(0, 0, 0, l[0]);
  } else {
    const int kc = min(key, L - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[ks] = *(const u32x4*)(pq(Uc) + (int64_t)kc * ldq + 2 * H * HD + (2 * ks + g) * 8);
    wait_vm(2 * rec_ops);
    block_sync();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const u32x4*)(Kb + wave * 4096 + fo(ks));
  }
  // ---------------------------------------------------------------- steps
  // service wave: dQ rows [tx * 32, +32) of unit U from the seven dS images of parity par and the K tile kpar
  auto dq_tile = [&](const Unit& U, int tx, int par, int kpar) {
    const char* img0 = Im + par * (NT * IMG);
    const char* tK = Kb + kpar * KB;
    f32x16 acc[2][2];   // [d half][cc]: four independent MFMA chains, added at the end
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // software pipeline over the key tiles: the operands of key tile kt+1 are read while the MFMAs of key tile kt run; the scheduling
    // fences keep it at two stages (fully unrolled and unfenced, hipcc hoists all 84 transpose reads and spills)
    u32x4 av[2][2][2], bv[2][2];   // [stage][cc][dt], [stage][cc]
    auto load_stage = [&](int kt, int st) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const u32x2 lo = tr8(img0 + kt * IMG + ir(cc, 0)), hi = tr8(img0 + kt * IMG + ir(cc, 1));
        const uint32_t lx = lo.x, ly = lo.y, hx = hi.x, hy = hi.y;
        bv[st][cc] = mk4(lx, ly, hx, hy);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) av[st][cc][dt] = tr_chunk(tK, kt * 32 + 16 * cc, dt);
      }
    };
    load_stage(0, 0);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (kt + 1 < NT) load_stage(kt + 1, (kt + 1) & 1);
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma_chunk<T>(acc[dt][cc], av[kt & 1][cc][dt], bv[kt & 1][cc]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int q = tx * 32 + ql;
    T* gq = pg(U);
    if (q < L) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const uint32_t lo = pack2(acc[dt][0][4 * rq] + acc[dt][1][4 * rq], acc[dt][0][4 * rq + 1] + acc[dt][1][4 * rq + 1], (T*)0);
          const uint32_t hi = pack2(acc[dt][0][4 * rq + 2] + acc[dt][1][4 * rq + 2], acc[dt][0][4 * rq + 3] + acc[dt][1][4 * rq + 3], (T*)0);
          __builtin_nontemporal_store(mk2(lo, hi), (u32x2*)(gq + (int64_t)q * ldq + dt * 32 + 8 * rq + 4 * g));
        }
    }
  };
  int slot_use = 0, slot_new = NS - 1;   // ring slots of record s (read by the pairs) and of record s + NS - 1 (filled now); both advance mod NS
  // start of step s (position t in its unit): every wave waits for its own copies of record s+1 (at t == 6: of the next unit's K too),
  // then the barrier makes them visible and retires step s-1; then this wave's copies for record s + NS - 1 (and K at t = 1..4)
  auto step_begin = [&](int s, int t, int ui) {
    const int tm1 = t == 0 ? 6 : t - 1, tm2 = tm1 == 0 ? 6 : tm1 - 1;
    if (s >= 80 && s <= 82) ALPRO_TS((s - 80) * 5);       // measurement build: steps 80..82 of the stamped workgroup, 5 stamps each
    if (wave == 7) {
      // service wave: it consumes its plain loads (lse, touches) at the end of every step, and loads return in order, so its one copy per
      // step has landed by then as well (an L2 hit, thanks to the touches three steps earlier)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
      wait_vm(ops_at(tm1) + (t == 6 ? 0 : ops_at(tm2)));
    if (s >= 80 && s <= 82) ALPRO_TS((s - 80) * 5 + 1);
    block_sync();
    fresh();
    if (s >= 80 && s <= 82) ALPRO_TS((s - 80) * 5 + 2);
  };
  // this wave's copies of the step, one at a time (j = 0, 1: record pieces, 2: the K piece at t = 1..4).  A copy holds its wave at issue
  // for as long as the CU's copy path needs for the 1 KiB pieces queued before it (~80 cycles each: all 8 waves issuing right behind the
  // barrier stood still for 1500-2400 cycles per step), so the pair waves spread theirs between their MFMA blocks, the lower and the upper
  // wave of a SIMD at different points.
  auto step_copy = [&](int t, int ui, int j) {
    if (j == 2) {
      if (t >= 1 && t <= 4) issue_k(Un, (ui + 1) & 1, t - 1);
    } else if (t + NS - 1 < NT) issue_record_op(Uc, t + NS - 1, slot_new, j);
    else issue_record_op(Un, t + NS - 1 - NT, slot_new, j);
  };
  auto next_step = [&] {
    slot_use = slot_use + 1 == NS ? 0 : slot_use + 1;
    slot_new = slot_new + 1 == NS ? 0 : slot_new + 1;
  };
  auto next_unit = [&](int ui) {
    Up = Uc;
    Uc = Un;
    Un = Unn;
    Unn = unit_at(ui + 3);
  };
  // Two loops with the same barrier sequence, so that what the service wave carries from step to step (touches in flight, lse queue) is
  // not live in the pair waves' loop, and the pair state (dK^T / dV^T accumulators, K / V fragments: 96 registers) not in the service loop.
  if (wave == 7) {
    // ================================================================ service wave
    // L2 warm-up: one dword of every 128-byte line the copies of the coming steps will ask for, 3 steps before they do -- the copy engine
    // holds few requests, and at HBM latency they turn over slowly (profiles/r3_attn_bwd_phase_stamps.txt).  Per step: lanes 0-31 row ql
    // of record s+7 (tile t of the next unit) in Q / dO / O; the other lanes rows of the next units' K / V by step position.  A touch is
    // consumed (its register released) three steps after its issue.
    uint32_t tq[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) tq[i][j] = 0;
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        const int s = ui * NT + t;
        step_begin(s, t, ui);
        step_copy(t, ui, 0);   // (the service wave has one piece per step and no K piece)
        const int t4 = t + NS - 1;
        const float lse_in = pl(t4 < NT ? Uc : Un)[min((t4 < NT ? t4 : t4 - NT) * 32 + ql, L - 1)];
        asm volatile("" ::"v"(tq[0][0]), "v"(tq[0][1]), "v"(tq[0][2]), "v"(tq[0][3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tq[0][j] = tq[1][j];
          tq[1][j] = tq[2][j];
        }
        if (flags & 1) {
          const int64_t rrow = min(t * 32 + ql, L - 1);
          const int64_t vrow = min(192 + ql, L - 1);
          const int64_t krow = min((t == 0 ? 192 : (t <= 3 ? (t - 1) * 64 : (t - 4) * 64)) + lane, L - 1);
          const T* qn = pq(Un);
          const T* kv = t == 0 ? qn + H * HD : (t <= 3 ? qn + 2 * H * HD : pq(Unn) + H * HD);   // K(next) tail | V(next) | K(next+1)
          tq[2][0] = *(const uint32_t*)(g == 0 ? qn + rrow * ldq : qn + vrow * ldq + 2 * H * HD);   // Q rows | V(next) tail
          tq[2][1] = *(const uint32_t*)(pd(Un) + rrow * ldo);
          tq[2][2] = *(const uint32_t*)(po(Un) + rrow * ldo);
          tq[2][3] = *(const uint32_t*)(kv + krow * ldq);
        }
        if (s > 0) {
          if (t == 0) dq_tile(Up, NT - 1, (s - 1) & 1, (ui - 1) & 1);
          else dq_tile(Uc, t - 1, (s - 1) & 1, ui & 1);
        }
        if (s + 1 < S) {
          const int sl1 = slot_use + 1 == NS ? 0 : slot_use + 1;
          stats(sl1, t + 1 == NT ? 0 : t + 1, (s + 1) & 1, Lq[((s + 1) & 7) * 32 + ql]);
        }
        if (g == 0) Lq[((s + NS - 1) & 7) * 32 + ql] = lse_in;
        if (s >= 80 && s <= 82) ALPRO_TS((s - 80) * 5 + 4);
        next_step();
      }
      next_unit(ui);
    }
    step_begin(S, 0, nu);   // step S: the last dQ tile
    step_copy(0, nu, 0);
    dq_tile(Up, NT - 1, (S - 1) & 1, (nu - 1) & 1);
    asm volatile("" ::"v"(tq[0][0]), "v"(tq[0][1]), "v"(tq[0][2]), "v"(tq[0][3]), "v"(tq[1][0]), "v"(tq[1][1]), "v"(tq[1][2]), "v"(tq[1][3]),
                 "v"(tq[2][0]), "v"(tq[2][1]), "v"(tq[2][2]), "v"(tq[2][3]));
  } else {
    // ================================================================ pair waves: (query tile s, key tile `wave`)
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
      f32x16 dk[2], dv[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        const int s = ui * NT + t;
        step_begin(s, t, ui);
        const char* rec = Rg + slot_use * REC;
        const char* tQ = rec;
        const char* tD = rec + 4096;
        const float* st = St + (s & 1) * 64;
        char* img = Im + ((s & 1) * NT + wave) * IMG;
        f32x16 sc, dp;
        float si = s_init;
        asm volatile("" : "+v"(si));   // (not hoisted as a 16-register splat, which then spills)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] = si;
          dp[r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 qa = *(const u32x4*)(tQ + fo(ks));
          const u32x4 da = *(const u32x4*)(tD + fo(ks));
          mma_chunk<T>(sc, qa, kf[ks]);
          mma_chunk<T>(dp, da, vf[ks]);
        }
        if (wave < 4) step_copy(t, ui, 0);
        if (t == NT - 1 && ui + 1 < nu) {   // kf / vf are dead for this unit: fetch the next unit's
          const int kc = min(key, L - 1);
          const char* tKn = Kb + ((ui + 1) & 1) * KB;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            vf[ks] = *(const u32x4*)(pq(Un) + (int64_t)kc * ldq + 2 * H * HD + (2 * ks + g) * 8);
            kf[ks] = *(const u32x4*)(tKn + wave * 4096 + fo(ks));
          }
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 lq = *(const float4*)(st + 8 * rq + 4 * g);
          const float4 dq4 = *(const float4*)(st + 32 + 8 * rq + 4 * g);
          const float ll[4] = {lq.x, lq.y, lq.z, lq.w};
          const float dd[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const float p = __builtin_amdgcn_exp2f(fmaf(sc[r], sl, ll[e]));
            sc[r] = p;
            dp[r] = p * fmaf(dp[r], scale, dd[e]);  // dS
          }
        }
        if (wave >= 4) step_copy(t, ui, 0);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          float pv[8], sv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pv[e] = sc[cc * 8 + e];
            sv[e] = dp[cc * 8 + e];
          }
          const u32x4 pb = pack_chunk<T>(pv), sb = pack_chunk<T>(sv);
          *(u32x2*)(img + iw(cc, 0)) = mk2(sb.x, sb.y);
          *(u32x2*)(img + iw(cc, 1)) = mk2(sb.z, sb.w);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            mma_chunk<T>(dv[dt], tr_chunk(tD, 16 * cc, dt), pb);
            mma_chunk<T>(dk[dt], tr_chunk(tQ, 16 * cc, dt), sb);
          }
          if (cc == 0) {
            if (wave < 4) step_copy(t, ui, 1);
            else step_copy(t, ui, 2);
          }
        }
        if (wave < 4) step_copy(t, ui, 2);
        if (s >= 80 && s <= 82) ALPRO_TS((s - 80) * 5 + 4);
        next_step();
      }
      if (key < L) {
        T* gk = pg(Uc) + (int64_t)key * ldq;
        store_row64<T>(gk + H * HD, dk, lane);
        store_row64<T>(gk + 2 * H * HD, dv, lane);
      }
      next_unit(ui);
    }
    step_begin(S, 0, nu);   // step S: nothing left to pair
    step_copy(0, nu, 0);
    step_copy(0, nu, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeats issued by the last steps
}

#endif  // ALPRO_ABLATIONS (persistent key-owned backward)
// ================================================================================================
// 16-bit temporal-attention backward: one WAVE per (32 consecutive tokens, head) unit, everything wave-private.
// The four 4 KiB tiles K, V, Q, dO of the unit go global -> LDS by DMA (16 copies per unit, swizzled on the source side) and
// every operand is then read from LDS; delta = rowsum(P o dP) (== rowsum(dO o O) for the recomputed P), so the saved
// output is not read at all; dQ / dK / dV leave through the dead K / V tiles as 16-byte row stores.  No workgroup
// barrier anywhere: 4 independent waves per workgroup, 2 workgroups per CU, units handed out grid-stride.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_temporal_bwd16_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                    const float* __restrict__ lse, T* __restrict__ dqkv, int64_t rows, int Tn,
                                                                    int H, float scale, int64_t units) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int WB = 4 * 4096 + 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* base = smem + wave * WB;
  char* tK = base;
  char* tV = base + 4096;
  char* tQ = base + 8192;
  char* tD = base + 12288;
  float* Ls = (float*)(base + 16384);  // -lse * log2(e); -inf on padded queries
  float* Ds = Ls + 32;                  // -delta * scale
  const uint32_t lds0 = lds_addr_of(base);
  const char* zero = (const char*)g_bwd_zero;
  const int64_t ldq = 3 * (int64_t)H * HD, ldo = (int64_t)H * HD;
  const int g = lane >> 5, ql = lane & 31;
  const float sl = scale * LOG2E_B;
  const int qgrp = ql / Tn;
  for (int64_t unit = (int64_t)blockIdx.x * 4 + wave; unit < units; unit += (int64_t)gridDim.x * 4) {
    const int64_t chunk = unit / H;
    const int h = (int)(unit - chunk * H);
    const int64_t r0 = chunk * 32;
    const int Le = (int)((rows - r0) < 32 ? (rows - r0) : 32);
    const T* qb = qkv + r0 * ldq + h * HD;
    const T* dob = dout + r0 * ldo + h * HD;
    T* db = dqkv + r0 * ldq + h * HD;
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) {
      const int row = piece * 8 + (lane >> 3), slot = lane & 7;
      const int ch = slot ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
      const bool ok = row < Le;
      const T* src = qb + (int64_t)row * ldq + ch * 8;
      dma16(ok ? (const char*)(src + H * HD) : zero, __builtin_amdgcn_readfirstlane(lds0 + piece * 1024));
      dma16(ok ? (const char*)(src + 2 * H * HD) : zero, __builtin_amdgcn_readfirstlane(lds0 + 4096 + piece * 1024));
      dma16(ok ? (const char*)src : zero, __builtin_amdgcn_readfirstlane(lds0 + 8192 + piece * 1024));
      dma16(ok ? (const char*)(dob + (int64_t)row * ldo + ch * 8) : zero, __builtin_amdgcn_readfirstlane(lds0 + 12288 + piece * 1024));
    }
    if (lane < 32) Ls[lane] = lane < Le ? -lse[unit * 32 + lane] * LOG2E_B : -INFINITY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ------------------------------------------------------------ phase 1: dQ (lane = query)
    u32x4 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const u32x4*)(tQ + tile_off<T>(ql, 2 * ks + g));
      dof[ks] = *(const u32x4*)(tD + tile_off<T>(ql, 2 * ks + g));
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      mma_chunk<T>(s, *(const u32x4*)(tK + tile_off<T>(ql, 2 * ks + g)), qf[ks]);
      mma_chunk<T>(dp, *(const u32x4*)(tV + tile_off<T>(ql, 2 * ks + g)), dof[ks]);
    }
    const float nlq = Ls[ql];
    float delta = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * g;
      const float p = (key / Tn == qgrp) ? __builtin_amdgcn_exp2f(fmaf(s[r], sl, nlq)) : 0.f;
      s[r] = p;
      delta = fmaf(p, dp[r], delta);
    }
    delta += __shfl_xor(delta, 32, 64);
    const float nds = -delta * scale;
    if (g == 0) Ds[ql] = nds;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] *= fmaf(dp[r], scale, nds);  // dS^T
    f32x16 acc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = s[cc * 8 + e];
      const u32x4 bop = pack_chunk<T>(v);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) mma_chunk<T>(acc[dt], load_t_chunk<T>(tK, 0, cc, lane, dt), bop);
    }
    // K / V rows of this lane's key as phase-2 B operands, then the K tile is dead: dQ leaves through it
    u32x4 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *(const u32x4*)(tK + tile_off<T>(ql, 2 * ks + g));
      vf[ks] = *(const u32x4*)(tV + tile_off<T>(ql, 2 * ks + g));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    store_rows_via_lds<T>(tK, acc, db, ldq, 0, Le, lane);
    // ------------------------------------------------------------ phase 2: dK, dV (lane = key)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      mma_chunk<T>(s, *(const u32x4*)(tQ + tile_off<T>(ql, 2 * ks + g)), kf[ks]);
      mma_chunk<T>(dp, *(const u32x4*)(tD + tile_off<T>(ql, 2 * ks + g)), vf[ks]);
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float4 lq = *(const float4*)(Ls + 8 * rq + 4 * g);
      const float4 dq4 = *(const float4*)(Ds + 8 * rq + 4 * g);
      const float ll[4] = {lq.x, lq.y, lq.z, lq.w}, dd[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * rq + e;
        const float p = ((8 * rq + 4 * g + e) / Tn == qgrp) ? __builtin_amdgcn_exp2f(fmaf(s[r], sl, ll[e])) : 0.f;
        s[r] = p;
        dp[r] = p * fmaf(dp[r], scale, dd[e]);
      }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      float pv[8], sv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pv[e] = s[cc * 8 + e];
        sv[e] = dp[cc * 8 + e];
      }
      const u32x4 pb = pack_chunk<T>(pv), sb = pack_chunk<T>(sv);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        mma_chunk<T>(dv[dt], load_t_chunk<T>(tD, 0, cc, lane, dt), pb);
        mma_chunk<T>(dk[dt], load_t_chunk<T>(tQ, 0, cc, lane, dt), sb);
      }
    }
    store_rows_via_lds<T>(tK, dk, db + H * HD, ldq, 0, Le, lane);
    store_rows_via_lds<T>(tV, dv, db + 2 * H * HD, ldq, 0, Le, lane);
  }
}

template <typename T, int NKT, bool HAS_BIAS, bool DROP>
int launch_bwd16(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int batch, int L, int H, float scale,
                 const float* key_bias, float dp, uint32_t ds, hipStream_t st) {
  const size_t lds = 2 * (size_t)NKT * 32 * 128 + 4 * (NKT == 8 ? 2048 : 4096) + 3 * (size_t)NKT * 32 * sizeof(float);
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)attn_bwd16_kernel<T, NKT, HAS_BIAS, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL((attn_bwd16_kernel<T, NKT, HAS_BIAS, DROP>), dim3((unsigned)(batch * H)), dim3(256), lds, st, (const T*)qkv, (const T*)out, (const T*)dout,
                     lse, (T*)dqkv, L, H, scale, key_bias, dp, ds, get_option(OPT_ATTN_ORDER));
  return check_launch("alpro_attn_bwd");
}

template <typename T, int NKT, bool DROP>
int launch_bwd16k(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int batch, int L, int H, float scale,
                  const float* key_bias, float dp, uint32_t ds, hipStream_t st) {
  constexpr int LP = NKT * 32;
  const size_t lds = 3 * (size_t)LP * 128 + (size_t)(NKT <= 7 ? 4 : 3) * NKT * 2048 + 3 * (size_t)LP * sizeof(float);
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)attn_bwd16k_kernel<T, NKT, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL((attn_bwd16k_kernel<T, NKT, DROP>), dim3((unsigned)(batch * H)), dim3(512), lds, st, (const T*)qkv, (const T*)out, (const T*)dout,
                     lse, (T*)dqkv, L, H, scale, key_bias, dp, ds, get_option(OPT_ATTN_ORDER));
  return check_launch("alpro_attn_bwd");
}

#ifdef ALPRO_ABLATIONS
template <typename T>
int launch_bwd16p(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int batch, int L, int H, float scale,
                  int flags, hipStream_t st) {
  const size_t lds = 2 * 7 * 4096 + 2 * 7 * 2048 + 5 * 3 * 4096 + (2 * 64 + 8 * 32) * sizeof(float);
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)attn_bwd16p_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int units = batch * H;
  hipLaunchKernelGGL((attn_bwd16p_kernel<T>), dim3((unsigned)(units < 256 ? units : 256)), dim3(512), lds, st, (const T*)qkv, (const T*)out,
                     (const T*)dout, lse, (T*)dqkv, L, H, scale, units, flags);
  return check_launch("alpro_attn_bwd");
}
#endif

template <typename T, int NKT, int NW, bool GROUPED>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int64_t nblocks_b, int L, int H, float scale,
               const float* key_bias, int Tn, int64_t total_rows, float dp, uint32_t ds, hipStream_t st) {
  const size_t lds = 2 * (size_t)NKT * 32 * BCfg<T>::RB + 3 * (size_t)NKT * 32 * sizeof(float);
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T, NKT, NW, GROUPED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL((attn_bwd_kernel<T, NKT, NW, GROUPED>), dim3((unsigned)(nblocks_b * H)), dim3(NW * 64), lds, st, (const T*)qkv, (const T*)out,
                     (const T*)dout, lse, (T*)dqkv, L, H, scale, key_bias, Tn, total_rows, dp, ds);
  return check_launch("alpro_attn_bwd");
}

template <typename T>
int dispatch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int batch, int L, int H, float scale,
                 const float* key_bias, float dp, uint32_t ds, hipStream_t st) {
  const int nkt = (L + 31) / 32;
  const int64_t rows = (int64_t)batch * L;
  if constexpr (sizeof(T) == 2) {
#define ALPRO_BWD16(N)                                                                                                          \
  do {                                                                                                                          \
    if (ds)                                                                                                                     \
      return key_bias ? launch_bwd16<T, N, true, true>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st)     \
                      : launch_bwd16<T, N, false, true>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st);   \
    return key_bias ? launch_bwd16<T, N, true, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st)      \
                    : launch_bwd16<T, N, false, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st);    \
  } while (0)
    if (nkt <= 2) ALPRO_BWD16(2);
    if (nkt <= 4) ALPRO_BWD16(4);
    // attn_bwd option: 0 two-phase everywhere; 1 (default) the measured best per shape: key-owned with 8 key tiles (fusion encoder,
    // L = 237: 0.48 vs 0.54 ms at 256 sequences), two-phase below (ViT spatial L = 197: 0.55 vs 0.63 ms at 512 sequences -- one 8-wave
    // workgroup per CU leaves its input latency uncovered); 2 key-owned wherever it applies (>= 5 key tiles); 3 / 4 the persistent
    // key-owned kernel with / without its L2 touches where IT applies (7 key tiles, no bias, no dropout), else as 2.
    const int kind = get_option(OPT_ATTN_BWD);
#ifdef ALPRO_ABLATIONS   // (the product library refuses attn_bwd 3 / 4: core.hip option_allowed)
    if (nkt == 7 && !key_bias && !ds && kind >= 3)
      return launch_bwd16p<T>(qkv, out, dout, lse, dqkv, batch, L, H, scale, kind == 3 ? 1 : 0, st);
#endif
    if ((nkt >= 5 && kind >= 2) || (nkt == 8 && kind == 1)) {   // every (query tile, key tile) pair once, one 8-wave workgroup per CU
#define ALPRO_BWD16K(N)                                                                                        \
  return ds ? launch_bwd16k<T, N, true>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st)   \
            : launch_bwd16k<T, N, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st)
      if (nkt <= 7) ALPRO_BWD16K(7);
      ALPRO_BWD16K(8);
#undef ALPRO_BWD16K
    }
    if (nkt <= 7) ALPRO_BWD16(7);
    ALPRO_BWD16(8);
#undef ALPRO_BWD16
  }
  if (nkt <= 2) return launch_bwd<T, 2, 4, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, 0, rows, dp, ds, st);
  if (nkt <= 4) return launch_bwd<T, 4, 4, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, 0, rows, dp, ds, st);
  if (nkt <= 7) return launch_bwd<T, 7, 4, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, 0, rows, dp, ds, st);
  return launch_bwd<T, 8, 4, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, 0, rows, dp, ds, st);
}

}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int dtype, int batch, int L,
                              int H, float scale, const float* key_bias, float drop_p, uint32_t drop_seed, void* stream) {
  ALPRO_CHECK(qkv && out && dout && lse && dqkv && batch > 0 && H > 0, "alpro_attn_bwd: bad args");
  ALPRO_CHECK(L > 0 && L <= 256, "alpro_attn_bwd: L=%d unsupported (1..256)", L);
  ALPRO_DISPATCH_DTYPE(dtype, T, return dispatch_bwd<T>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, drop_p, drop_seed, (hipStream_t)stream));
  return ALPRO_OK;
}

extern "C" int alpro_attn_temporal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int dtype,
                                       int64_t rows, int T, int H, float scale, void* stream) {
  ALPRO_CHECK(qkv && out && dout && lse && dqkv && rows > 0 && H > 0, "alpro_attn_temporal_bwd: bad args");
  ALPRO_CHECK(T > 0 && 32 % T == 0 && rows % T == 0, "alpro_attn_temporal_bwd: num_frm=%d must divide 32 and rows", T);
  const int64_t chunks = (rows + 31) / 32;
  if (dtype != ALPRO_F32) {
    const int64_t units = chunks * H;
    int64_t grid = (units + 3) / 4;
    if (grid > 512) grid = 512;
    const size_t lds = 4 * (4 * 4096 + 256);
    static DeviceOnce attr_once;
    attr_once.run([&] {
      (void)hipFuncSetAttribute((const void*)attn_temporal_bwd16_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)attn_temporal_bwd16_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (dtype == ALPRO_BF16) {
      hipLaunchKernelGGL(attn_temporal_bwd16_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                         (bf16_t*)dqkv, rows, T, H, scale, units);
    } else {
      hipLaunchKernelGGL(attn_temporal_bwd16_kernel<f16_t>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, (const f16_t*)qkv, (const f16_t*)dout, lse,
                         (f16_t*)dqkv, rows, T, H, scale, units);
    }
    return check_launch("alpro_attn_temporal_bwd");
  }
  ALPRO_DISPATCH_DTYPE(dtype, T_, return (launch_bwd<T_, 1, 1, true>(qkv, out, dout, lse, dqkv, chunks, 32, H, scale, nullptr, T, rows, 0.f, 0u, (hipStream_t)stream)));
  return ALPRO_OK;
}

#ifdef ALPRO_ABLATIONS
// measurement build only: choose the workgroup whose phases are stamped (block < 0: none) / read the 8 x 16 stamps back
extern "C" int alpro_debug_attn_bwd_stamps(int block, unsigned long long* out128) {
  if (out128) {
    if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_attn_ts), sizeof(unsigned long long) * 128) != hipSuccess) return ALPRO_ERR_LAUNCH;
  }
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_ts_block), &block, sizeof(int)) != hipSuccess) return ALPRO_ERR_LAUNCH;
  return ALPRO_OK;
}
extern "C" int alpro_debug_attn_bwd_alias(int n) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_alias), &n, sizeof(int)) == hipSuccess ? ALPRO_OK : ALPRO_ERR_LAUNCH;
}
#endif
