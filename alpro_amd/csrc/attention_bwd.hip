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

template <typename T, int NKT, bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_bwd16_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                          const T* __restrict__ dout, const float* __restrict__ lse,
                                                                          T* __restrict__ dqkv, int L, int H, float scale,
                                                                          const float* __restrict__ key_bias, float drop_p, uint32_t drop_seed) {
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
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
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
            if (drop_seed) gd = drop_keep(drop_seed, (dbase + qc) * L + kt * 32 + 8 * rq + 4 * g + e, dth) ? gd * dks : 0.f;
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
          if (drop_seed) {
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

template <typename T, int NKT, bool HAS_BIAS>
int launch_bwd16(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int batch, int L, int H, float scale,
                 const float* key_bias, float dp, uint32_t ds, hipStream_t st) {
  const size_t lds = 2 * (size_t)NKT * 32 * 128 + 4 * (NKT == 8 ? 2048 : 4096) + 3 * (size_t)NKT * 32 * sizeof(float);
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)attn_bwd16_kernel<T, NKT, HAS_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL((attn_bwd16_kernel<T, NKT, HAS_BIAS>), dim3((unsigned)(batch * H)), dim3(256), lds, st, (const T*)qkv, (const T*)out, (const T*)dout,
                     lse, (T*)dqkv, L, H, scale, key_bias, dp, ds);
  return check_launch("alpro_attn_bwd");
}

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
#define ALPRO_BWD16(N)                                                                                              \
  return key_bias ? launch_bwd16<T, N, true>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st)   \
                  : launch_bwd16<T, N, false>(qkv, out, dout, lse, dqkv, batch, L, H, scale, key_bias, dp, ds, st)
    if (nkt <= 2) ALPRO_BWD16(2);
    if (nkt <= 4) ALPRO_BWD16(4);
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
