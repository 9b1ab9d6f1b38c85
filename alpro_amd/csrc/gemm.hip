// NT GEMM with fused epilogue for gfx950:  C = epilogue(A[M,K] * W[N,K]^T).
//
// Tile 128x128 per 256-thread workgroup (4 waves, 2x2, each 64x64 = 2x2 MFMA 32x32 accumulators),
// K-tile = 128 BYTES per row (64 bf16/f16 or 32 f32), so staging, LDS image and fragment reads are
// identical for every storage dtype; only mma_chunk<T> differs (common.hpp).
//  * global -> registers -> LDS double buffer, next tile's loads issued before the MFMAs of the
//    current one (one barrier per K-tile);
//  * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row >> 1) & 7 so that every
//    ds_read_b128 lane group of a fragment read hits 16 distinct 16-B slots of the 256-B bank row;
//  * workgroup -> tile map is XCD-aware: each XCD (blockIdx % 8) walks a contiguous range of tiles
//    with the N tiles of one A row-panel adjacent, so the panel is fetched into one L2 only.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace alpro {

namespace {
constexpr int BM = 128, BN = 128, ROWB = 128, NT = 256;
constexpr int TILE_BYTES = BM * ROWB;  // 16 KiB per operand per buffer

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct RowDst {
  int64_t out, res;
  bool side;
};
template <int mode>
__device__ __forceinline__ RowDst map_row(int p0, int p1, int m) {
  RowDst d;
  d.side = false;
  if (mode == ALPRO_MAP_IDENTITY) {
    d.out = d.res = m;
  } else if (mode == ALPRO_MAP_SKIP_CLS) {
    d.out = d.res = (int64_t)m + m / p0 + 1;
  } else if (mode == ALPRO_MAP_FRAME_TOKENS) {
    const int T = p0, N = p1;
    const int bt = m / (N + 1), j = m - bt * (N + 1);
    const int b = bt / T, t = bt - b * T;
    if (j == 0) {
      d.side = true;
      d.out = bt;
      d.res = -1;
    } else {
      d.out = d.res = (int64_t)b * (1 + N * T) + 1 + (int64_t)(j - 1) * T + t;
    }
  } else {  // PATCH_EMBED
    const int T = p0, N = p1;
    const int bt = m / N, n = m - bt * N;
    const int b = bt / T, t = bt - b * T;
    d.out = (int64_t)b * (1 + N * T) + 1 + (int64_t)n * T + t;
    d.res = (int64_t)n * T + t;
  }
  return d;
}

template <typename T>
__device__ __forceinline__ void store_c(void* C, int c_dtype, int64_t idx, float v) {
  if (c_dtype == ALPRO_F32) ((float*)C)[idx] = v;
  else ((T*)C)[idx] = from_f32<T>(v);
}


// Wave-private LDS hand-off: DS operations of one wave execute in issue order, so a ds_read after a ds_write of the
// same wave needs no hardware wait -- only the compiler must not reorder them.
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// The same hand-off without the wait: pins the COMPILER's issue order only (no instruction).  The compiler reasons per lane -- when it can
// prove that a lane's own staging writes and its own reads never overlap, it may move the read above a write that ANOTHER lane's read depends
// on.  Round 4 hit exactly that: the fp32-output epilogue of the 8-phase kernel read its first staged row before the last ds_write2 of the
// fragment row had been issued (256 stale elements per tile, tests/test_hip_ops.py::test_gemm_8phase_kernel[f32res]); the other staging
// epilogues had been in source order by luck.  Every stage write block is now bracketed by this.
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }

template <typename T, int ACT> __device__ __forceinline__ float apply_act(float x) {
  if (ACT == ALPRO_ACT_GELU) return gelu_fast<T>(x);   // (GELU_SAVE_GRAD computes gelu together with gelu' before this point)
  if (ACT == ALPRO_ACT_RELU) return fmaxf(x, 0.f);
  return x;
}

// Output / residual accesses are non-temporal: they are streamed once (150-600 MB per launch against 32 MB of L2), and
// keeping them out of the L2 allocation path is worth 7-8 % on the bf16-output GEMMs (round-2 measurement).
// Epilogue of 16 staged rows x 64 columns of one wave: lane l handles columns 4*(l&15)..+3 of rows p*4 + (l>>4),
// p = 0..3, so every global access is a 16-byte (fp32) / 8-byte (16-bit) piece of a 256-/128-byte row segment.
// FAST (wave-uniform): the whole 16x64 block is in range and every stride is vector-aligned -> no per-element
// predication at all (the predicated variant is ~4x the instructions and was costing ~11 us per 256x256 tile).
template <typename T, int ACT, int MAP, bool FAST, int PASSES = 4>
__device__ __forceinline__ void epi_rows16(const alpro_gemm_desc_t& g, const float* stage, int m_base, int n_base, int lane, const float (&bias)[4],
                                           const float4* pre_res = nullptr) {
  const int c4 = (lane & 15) * 4;
  const int n = n_base + c4;
  float4 rr[PASSES];
  int64_t orow[PASSES];
  bool live[PASSES], side[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int m = m_base + p * 4 + (lane >> 4);
    live[p] = FAST || (m < g.M && n < g.N);
    const RowDst d = map_row<MAP>(g.map_p0, g.map_p1, live[p] ? m : 0);
    orow[p] = d.out;
    side[p] = d.side;
    rr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pre_res) {
      rr[p] = pre_res[p];  // already in flight / landed: issued two chunks ago by the caller
    } else if (g.residual && live[p] && !d.side) {
      const float* rp = g.residual + d.res * g.ldr + n;
      if (FAST) {
        const f32x4 t = __builtin_nontemporal_load((const f32x4*)rp);  // streamed once
        rr[p] = make_float4(t.x, t.y, t.z, t.w);
      }
      else {
        rr[p].x = rp[0];
        if (n + 1 < g.N) rr[p].y = rp[1];
        if (n + 2 < g.N) rr[p].z = rp[2];
        if (n + 3 < g.N) rr[p].w = rp[3];
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int row = p * 4 + (lane >> 4);
    if (!live[p]) continue;
    const float4 a = *(const float4*)(stage + row * 64 + c4);
    float v[4] = {a.x, a.y, a.z, a.w};
    const float res[4] = {rr[p].x, rr[p].y, rr[p].z, rr[p].w};
    const float rs = g.row_scale ? g.row_scale[(g.m_off + m_base + row) / g.row_scale_group] : 1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g.alpha * v[e] + bias[e];
    if ((ACT == ALPRO_ACT_GELU || ACT == ALPRO_ACT_RELU) && g.C2) {  // pre-activation copy (host guarantees vector alignment for C2)
      if constexpr (sizeof(T) == 2) {
        __builtin_nontemporal_store(mk2(pack2(v[0], v[1], (T*)0), pack2(v[2], v[3], (T*)0)), (u32x2*)((T*)g.C2 + orow[p] * g.ldc2 + n));
      } else {
        *(float4*)((float*)g.C2 + orow[p] * g.ldc2 + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    if (ACT == ALPRO_ACT_GELU_SAVE_GRAD) {  // v = gelu(v), C2 = gelu'(v)
      float dv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) gelu_and_grad<T>(v[e], v[e], dv[e]);
      if (FAST) {
        if constexpr (sizeof(T) == 2) {
          __builtin_nontemporal_store(mk2(pack2(dv[0], dv[1], (T*)0), pack2(dv[2], dv[3], (T*)0)), (u32x2*)((T*)g.C2 + orow[p] * g.ldc2 + n));
        } else {
          *(float4*)((float*)g.C2 + orow[p] * g.ldc2 + n) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < g.N) ((T*)g.C2)[orow[p] * g.ldc2 + n + e] = from_f32<T>(dv[e]);
      }
    }
    if (ACT == ALPRO_ACT_GELU_BWD || ACT == ALPRO_ACT_MUL_SAVED) {  // v *= gelu'(saved pre-activation) / v *= saved factor
      const T* pp = (const T*)g.C2 + orow[p] * g.ldc2 + n;
      float pre[4];
      if (FAST) {
        if constexpr (sizeof(T) == 2) {
          const u32x2 u = *(const u32x2*)pp;
          const uint32_t ux = u.x, uy = u.y;
          pre[0] = to_f32(T{(uint16_t)(ux & 0xFFFFu)});
          pre[1] = to_f32(T{(uint16_t)(ux >> 16)});
          pre[2] = to_f32(T{(uint16_t)(uy & 0xFFFFu)});
          pre[3] = to_f32(T{(uint16_t)(uy >> 16)});
        } else {
          const float4 f = *(const float4*)pp;
          pre[0] = f.x; pre[1] = f.y; pre[2] = f.z; pre[3] = f.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) pre[e] = (n + e < g.N) ? to_f32(pp[e]) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= (ACT == ALPRO_ACT_MUL_SAVED) ? pre[e] : gelu_grad<T>(pre[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act<T, ACT>(v[e]) * rs;
    if (MAP == ALPRO_MAP_IDENTITY && g.drop_seed) {
      const uint32_t th = drop_thresh24(g.drop_p);
      const float ks = 1.0f / (1.0f - g.drop_p);
      const uint64_t i0 = (uint64_t)(g.m_off + m_base + row) * (uint64_t)g.N + (uint64_t)n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = drop_keep(g.drop_seed, i0 + e, th) ? v[e] * ks : 0.f;
    }
    if constexpr (MAP == ALPRO_MAP_SKIP_CLS) {
      if (g.bias2) {  // unscaled second bias (merged temporal projection)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (FAST || n + e < g.N) ? g.bias2[n + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += res[e];
    if (MAP == ALPRO_MAP_FRAME_TOKENS && side[p]) {
      float* dst = g.side + orow[p] * g.ld_side + n;
      if (FAST) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < g.N) dst[e] = v[e];
      }
    } else if (FAST) {
      if (g.c_dtype == ALPRO_F32) {
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4*)((float*)g.C + orow[p] * g.ldc + n));
      } else if constexpr (sizeof(T) == 2) {
        __builtin_nontemporal_store(mk2(pack2(v[0], v[1], (T*)0), pack2(v[2], v[3], (T*)0)), (u32x2*)((T*)g.C + orow[p] * g.ldc + n));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < g.N) store_c<T>(g.C, g.c_dtype, orow[p] * g.ldc + n + e, v[e]);
    }
  }
}

// Residual rows of one 8-row chunk (the two passes of epi_rows16<.., PASSES = 2>) for a FAST tile.  The persistent kernel
// issues these one chunk ahead of their use (two would spill): loaded at the point of use, every chunk exposed a full HBM round trip
// (16 chunks x ~1.5 us = the whole 25 us epilogue of the N=768 fp32-residual GEMMs; 16 KiB in flight per CU = ~11 B/clk).
template <int MAP>
__device__ __forceinline__ void epi_prefetch_res(const alpro_gemm_desc_t& g, int m_base, int n_base, int lane, float4 (&rr)[2]) {
  const int n = n_base + (lane & 15) * 4;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const RowDst d = map_row<MAP>(g.map_p0, g.map_p1, m_base + p * 4 + (lane >> 4));
    rr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!d.side) {
      const f32x4 t = __builtin_nontemporal_load((const f32x4*)(g.residual + d.res * g.ldr + n));
      rr[p] = make_float4(t.x, t.y, t.z, t.w);
    }
  }
}

// 16-bit outputs under the identity map (qkv / proj / fc1 / every dgrad): 8 columns per lane -> one 16-byte store per
// lane, 8 rows per wave instruction.  The store path is ISSUE-bound per CU (~one wave-store per ~100 cycles measured),
// so halving the number of store instructions halves the epilogue tail.  Whole block in range (FAST) only.
// Round 5: no vector-memory load sits behind a run-time test inside a pass.  Rounds 3-4 tested g.row_scale / g.residual per pass; the row scale was
// a conditional per-lane load, and the join behind a conditional load is closed with s_waitcnt vmcnt(0): every one of a tile's 16 passes
// waited for the previous pass's output store to be acknowledged -- and, in the MUL_SAVED form, for the saved-factor rows fetched AHEAD, which
// defeated the run-ahead.  Now (i) the fp32 residual is a template parameter (RES: the caller tests the pointer once per tile), and (ii) the
// row scale of a pass comes from SCALAR loads: a pass covers 8 consecutive rows, which lie in at most two groups when row_scale_group >= 8
// (launcher: drop-path scales per 8-frame token group, per 197-token frame, per clip) -- one wave-uniform division, two s_load_dword, a
// compare per lane.  The plain GEMMs (qkv, fc1, dgrads) have no load at all in their passes: the stores stream.
template <typename T, int ACT, int PASSES = 2, int ABL = 0, bool RES = true>
__device__ __forceinline__ void epi_rows16_c16(const alpro_gemm_desc_t& g, const float* stage, int m_base, int n_base, int lane, const float (&bias)[8],
                                               const u32x4* pre_c2 = nullptr) {
  const int c8 = (lane & 7) * 8;
  const int n = n_base + c8;
  float rsv[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) rsv[p] = 1.0f;
  if (g.row_scale) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const uint32_t m0 = (uint32_t)g.m_off + (uint32_t)__builtin_amdgcn_readfirstlane(m_base) + p * 8;   // first row of the pass (wave-uniform; rows < 2^31)
      const uint32_t gi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(m0 / (uint32_t)g.row_scale_group));
      const uint32_t edge = (gi + 1) * (uint32_t)g.row_scale_group;
      const float lo = sload_f32(g.row_scale, gi), hi = sload_f32(g.row_scale, edge < (uint32_t)g.m_off + (uint32_t)g.M ? gi + 1 : gi);
      rsv[p] = (m0 + (uint32_t)(lane >> 3)) >= edge ? hi : lo;
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int row = p * 8 + (lane >> 3);
    const int64_t m = m_base + row;
    const float4 a0 = *(const float4*)(stage + row * 64 + c8), a1 = *(const float4*)(stage + row * 64 + c8 + 4);
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float rs = rsv[p];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g.alpha * v[e] + bias[e];
    if ((ACT == ALPRO_ACT_GELU || ACT == ALPRO_ACT_RELU) && g.C2) __builtin_nontemporal_store(pack_chunk<T>(v), (u32x4*)((T*)g.C2 + m * g.ldc2 + n));
    if (ACT == ALPRO_ACT_GELU_SAVE_GRAD) {
      float dv[8];
      if constexpr (sizeof(T) == 2) {   // 16-bit storage: the one-exponential form on pairs (common.hpp gelu_and_grad2)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2v yy, dd;
          gelu_and_grad2((f32x2v){v[e], v[e + 1]}, yy, dd);
          v[e] = yy.x; v[e + 1] = yy.y;
          dv[e] = dd.x; dv[e + 1] = dd.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) gelu_and_grad<T>(v[e], v[e], dv[e]);
      }
      __builtin_nontemporal_store(pack_chunk<T>(dv), (u32x4*)((T*)g.C2 + m * g.ldc2 + n));
    }
    if (ACT == ALPRO_ACT_GELU_BWD || ACT == ALPRO_ACT_MUL_SAVED) {
      float pre[8];
      unpack_chunk<T>(pre_c2 ? pre_c2[p] : *(const u32x4*)((const T*)g.C2 + m * g.ldc2 + n), pre);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= (ACT == ALPRO_ACT_MUL_SAVED) ? pre[e] : gelu_grad<T>(pre[e]);
    }
    if constexpr (ACT == ALPRO_ACT_GELU && sizeof(T) == 2) {   // the same arithmetic as gelu_fast, polynomial on pairs (v_pk_fma_f32)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2v yy = gelu_fast2((f32x2v){v[e], v[e + 1]});
        v[e] = yy.x * rs;
        v[e + 1] = yy.y * rs;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = apply_act<T, ACT>(v[e]) * rs;
    }
    if (g.drop_seed) {
      float dp = g.drop_p;
      asm volatile("" : "+s"(dp));   // keeps 1 / (1 - p) (and its packed-multiply splat) from being hoisted over the K loop as a kernel invariant, where it is spilled and reloaded per pass
      const uint32_t th = drop_thresh24(dp);
      const float ks = 1.0f / (1.0f - dp);
      const uint64_t i0 = (uint64_t)(g.m_off + m) * (uint64_t)g.N + (uint64_t)n;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = drop_keep(g.drop_seed, i0 + e, th) ? v[e] * ks : 0.f;
    }
    if constexpr (RES) {
      if (g.residual) {
        const f32x4 r0 = __builtin_nontemporal_load((const f32x4*)(g.residual + m * g.ldr + n)), r1 = __builtin_nontemporal_load((const f32x4*)(g.residual + m * g.ldr + n + 4));
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
    }
    if (ABL == 1) {  // ablation (gemm_tune 3): everything but the global store
      u32x4 keep = pack_chunk<T>(v);
      asm volatile("" ::"v"(keep));
    } else {
      __builtin_nontemporal_store(pack_chunk<T>(v), (u32x4*)((T*)g.C + m * g.ldc + n));
    }
  }
}

// wave-uniform test for the FAST epilogue of a (rows x 64) wave sub-tile
__device__ __forceinline__ bool epi_fast_ok(const alpro_gemm_desc_t& g, int m_base, int rows, int n_base) {
  return (m_base + rows <= g.M) && (n_base + 64 <= g.N) && ((g.ldc & 3) == 0) && (!g.residual || (g.ldr & 3) == 0) &&
         ((g.ld_side & 3) == 0);
}

__device__ __forceinline__ void load_bias4(const alpro_gemm_desc_t& g, int n, float (&bias)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = (g.bias && n + e < g.N) ? g.bias[n + e] : 0.f;
}

template <typename T, int ACT, int MAP>
__device__ __forceinline__ void gemm_nt_tile(const alpro_gemm_desc_t& g, const int bid, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
  const int nblk = ntn * ntm;
  if (bid >= nblk) return;  // (batched launches size the grid for the largest job)
  // XCD-aware, bijective remap of blockIdx -> logical tile
  int tile;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  const char* Ab = (const char*)g.A;
  const char* Wb = (const char*)g.W;
  const int64_t lda_b = g.lda * (int64_t)sizeof(T), ldw_b = g.ldw * (int64_t)sizeof(T);

  // staging: 1024 16-B chunks per operand tile, 4 per thread
  const char* a_src[4];
  const char* w_src[4];
  int st_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * NT, row = c >> 3, ch = c & 7;
    const int am = min(m0 + row, g.M - 1), wn = min(n0 + row, g.N - 1);
    a_src[i] = Ab + am * lda_b + ch * 16;
    w_src[i] = Wb + wn * ldw_b + ch * 16;
    st_off[i] = lds_off(row, ch);
  }
  // fragment read offsets (bytes within an operand tile) for k-step s: XOR of chunk index is per row
  int a_row[2], b_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_row[i] = wr * 64 + i * 32 + (lane & 31);
    b_row[i] = wc * 64 + i * 32 + (lane & 31);
  }
  const int khalf = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (g.K * (int)sizeof(T)) / ROWB;
  u32x4 ra[4], rw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *(const u32x4*)(a_src[i]);
    rw[i] = *(const u32x4*)(w_src[i]);
  }
  char* bufA = smem;
  char* bufW = smem + 2 * TILE_BYTES;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *(u32x4*)(bufA + st_off[i]) = ra[i];
    *(u32x4*)(bufW + st_off[i]) = rw[i];
  }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // issue the next tile's global loads before this tile's MFMAs (the last iteration re-reads its own,
    // L1-resident tile: keeping the loads unconditional keeps the staging registers out of scratch)
    {
      const int64_t ko = (int64_t)min(kt + 1, nk - 1) * ROWB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *(const u32x4*)(a_src[i] + ko);
        rw[i] = *(const u32x4*)(w_src[i] + ko);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // loads stay ahead of the MFMAs (hipcc would sink them to the ds_write)
    const char* cA = bufA + cur * TILE_BYTES;
    const char* cW = bufW + cur * TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *(const u32x4*)(cA + lds_off(a_row[i], 2 * s + khalf));
        fb[i] = *(const u32x4*)(cW + lds_off(b_row[i], 2 * s + khalf));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_chunk<T>(acc[i][j], fa[i], fb[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      char* nA = bufA + (cur ^ 1) * TILE_BYTES;
      char* nW = bufW + (cur ^ 1) * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(u32x4*)(nA + st_off[i]) = ra[i];
        *(u32x4*)(nW + st_off[i]) = rw[i];
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue (the trailing __syncthreads of the K loop guarantees nobody still reads the staging tiles)
  float* stage = (float*)(smem + wave * (64 * 64 * 4));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane), col = lane & 31;
    stage[row * 64 + col] = acc[0][0][r];
    stage[row * 64 + 32 + col] = acc[0][1][r];
    stage[(32 + row) * 64 + col] = acc[1][0][r];
    stage[(32 + row) * 64 + 32 + col] = acc[1][1][r];
  }
  wave_lds_sync();
  const int mb = m0 + wr * 64, nb = n0 + wc * 64;
  float bias[4];
  load_bias4(g, nb + (lane & 15) * 4, bias);
  if (epi_fast_ok(g, mb, 64, nb)) {
#pragma unroll
    for (int c = 0; c < 4; ++c) epi_rows16<T, ACT, MAP, true>(g, stage + c * 16 * 64, mb + c * 16, nb, lane, bias);
  } else {
#pragma unroll 1
    for (int c = 0; c < 4; ++c) epi_rows16<T, ACT, MAP, false>(g, stage + c * 16 * 64, mb + c * 16, nb, lane, bias);
  }
}

template <typename T, int ACT, int MAP>
__global__ __launch_bounds__(NT, 2) void gemm_nt_kernel(const alpro_gemm_desc_t g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_tile<T, ACT, MAP>(g, blockIdx.x, smem);
}

// Many small independent GEMMs in ONE launch (round 3): blockIdx.y = job, descriptors in device memory.  The merged temporal projection
// needs, per ViT block and optimizer step, W_e = W_fc W_p (a 768^3 product on 36 workgroups, 64 us) and, in backward, two more 768^3
// products for the product rule -- 12 blocks x 3 launches that each fill a seventh of the chip; batched they run side by side.
template <typename T, int ACT, int MAP>
__global__ __launch_bounds__(NT, 2) void gemm_nt_batch_kernel(const alpro_gemm_desc_t* __restrict__ descs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const alpro_gemm_desc_t g = descs[blockIdx.y];
  gemm_nt_tile<T, ACT, MAP>(g, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4, each 128x64 = 4x2 MFMA accumulators), K-tile 128 bytes, two LDS stages of
// 64 KiB filled by global_load_lds_dwordx4 (no VGPR round trip, no ds_write).  The LDS image written by
// the DMA is lane-linear (1 KiB = 8 rows per wave instruction), so the bank swizzle is applied to the
// per-lane SOURCE chunk and undone by the same XOR on the fragment read (linear dest + swizzled source).
constexpr int BM2 = 256, BN2 = 256, NT2 = 512;
constexpr int TILE2_BYTES = BM2 * ROWB;  // 32 KiB per operand per stage

// ------------------------------------------------------------------------------------------------
// Persistent form of the 256x256 kernel: one workgroup per CU walks its tiles (XCD-contiguous order).  On the
// K = 768 shapes of this model a tile is only 12 K-steps, so what the one-tile-per-workgroup kernel loses is the
// ~16 us per tile of workgroup turn-around + first-tile DMA latency + epilogue; here the first K-tile of the
// NEXT tile is DMA-prefetched before the epilogue runs, and the epilogue stages through its own 32 KiB of LDS
// (16 rows x 64 columns per wave at a time) so the two 64 KiB stage buffers are free to receive it.
// Fragment reads are register double-buffered (the reads of K-chunk s+1 are in flight under the MFMAs of s).
constexpr int EPI_BYTES = 8 * 16 * 64 * 4;  // 32 KiB: 8 waves x (16 rows x 64 cols) fp32

// TUNE: where the 8 DMA pieces of the next K-tile are issued among the 32 MFMAs of a K-step (experiment knob, ALPRO_GEMM_TUNE):
//   0  copy c after MFMA 4c+1 (waves 0-3) / 4c+3 (waves 4-7): spread over the whole step -- the last piece is issued ~100 cycles
//      before the step ends, so its full L2 / MALL latency is exposed at the next step's vmcnt(0)
//   1  copy c after MFMA 2c+1 / 2c+2: all pieces out in the first half of the step (default: +3-5 % on every shape,
//      round-2 A/B of the variants on the model shapes)
//   2  copy c after MFMA 3c+1 / 3c+2: first three quarters
//   3, 4  ablations of the 16-bit-output epilogue (no global stores / no epilogue at all; wrong results by construction) -- the
//      measurements and the three epilogue rewrites they led to are in profiles/r2_gemm_epilogue_experiments.txt
__device__ __forceinline__ constexpr int copy_slot(int tune, int q, int pos) {
  if (tune == 0) return ((q & 1) && ((q >> 1) & 1) == pos) ? (q >> 2) : -1;
  if (tune == 1 || tune >= 3) { const int r = q - 1 - pos; return (r >= 0 && r < 16 && (r & 1) == 0) ? (r >> 1) : -1; }
  const int r = q - 1 - pos;
  return (r >= 0 && r < 24 && r % 3 == 0) ? r / 3 : -1;
}

// Tail split (round 3): with nblk = Q * grid + R tiles, the last round keeps only R workgroups busy (M = 50176 x N = 768 at B = 32: 591
// tiles on 256 CUs = 2.31 -> 3 rounds, 77 %).  When 2R <= grid, each tile of that round is cut in two along M and handed to TWO
// workgroups: a half tile is a 128 x 256 tile whose upper wave row (waves 4-7, one per SIMD) idles -- it still issues its share of the
// DMA and takes the barriers -- so the round costs about half of a full one (2.31 -> 2.5 round-equivalents instead of 3).
template <typename T, int ACT, int MAP, int TUNE = 1>
__global__ __launch_bounds__(NT2, 2) void gemm_nt256p_kernel(const alpro_gemm_desc_t g, const int tail_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (g.N + BN2 - 1) / BN2, ntm = (g.M + BM2 - 1) / BM2;
  const int nblk = ntn * ntm;
  const int64_t lda_b = g.lda * (int64_t)sizeof(T), ldw_b = g.ldw * (int64_t)sizeof(T);
  const int nk = (g.K * (int)sizeof(T)) / ROWB;
  // XCD-contiguous walk: within one round of gridDim.x tiles, XCD x (= blockIdx % 8) owns a contiguous run
  const int per_xcd = (gridDim.x + 7) >> 3;
  const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;

  const char* a_src[4];
  const char* w_src[4];
  int m0 = 0, n0 = 0;
  const int G = gridDim.x;
  const int q_full = nblk / G, rem = nblk - q_full * G;
  const bool split = tail_split && rem > 0 && 2 * rem <= G;
  // it-th tile of this workgroup: (tile, half) with half = -1 for a full tile, 0 / 1 for the lower / upper 128 rows of a split tile
  auto locate = [&](int it, int& t, int& hf) -> bool {
    hf = -1;
    if (it < q_full) { t = slot + it * G; return true; }
    if (it > q_full) return false;
    if (split) {
      if (slot >= 2 * rem) return false;
      t = q_full * G + (slot >> 1);
      hf = slot & 1;
      return true;
    }
    t = q_full * G + slot;
    return slot < rem;
  };
  auto setup = [&](int tile, int hf) {
    const int tm = tile / ntn, tn = tile - tm * ntn;
    m0 = tm * BM2 + (hf > 0 ? BM2 / 2 : 0);
    n0 = tn * BN2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave + 8 * i) * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      a_src[i] = (const char*)g.A + min(m0 + row, g.M - 1) * lda_b + ch * 16;
      w_src[i] = (const char*)g.W + min(n0 + row, g.N - 1) * ldw_b + ch * 16;
    }
  };
  const uint32_t lds_base = lds_addr_of(smem);
  // copy c = 0..7 of K-tile kt into stage buffer buf: (A, W) x 4 pieces of 1 KiB per wave.  Issued from inline asm
  // (common.hpp dma16) and tracked by the hand-placed vmcnt waits below.
  auto copy_piece = [&](int c, int kt, int buf, bool half_a = false) {
    const int i = c >> 1;
    if (half_a && !(c & 1) && i >= 2) return;  // rows 128..255 of a half tile's A image are never read
    const char* src = ((c & 1) ? w_src[i] : a_src[i]) + (int64_t)kt * ROWB;
    dma16(src, __builtin_amdgcn_readfirstlane(lds_base + buf * 2 * TILE2_BYTES + (c & 1) * TILE2_BYTES + (wave + 8 * i) * 1024));
  };
  auto stage_tile = [&](int kt, int buf, bool half_a = false) {
#pragma unroll
    for (int c = 0; c < 8; ++c) copy_piece(c, kt, buf, half_a);
  };
  int a_row[4], b_row[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_row[i] = wr * 128 + i * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 2; ++j) b_row[j] = wc * 64 + j * 32 + (lane & 31);
  const int khalf = lane >> 5;
  float* stage = (float*)(smem + 4 * TILE2_BYTES + wave * (16 * 64 * 4));

  int it = 0, tile, hf;
  if (!locate(0, tile, hf)) return;
  // Invariant at the top of every tile: K-tiles 0 and 1 are in buffers s0 and s0^1 and this wave has no DMA in flight,
  // so the first two K-steps need no vmcnt wait -- the previous tile's output stores drain underneath them.
  const int pos = wave >> 2;
  auto wait_vm0 = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto block_sync = [] {  // barrier that does NOT drain vmcnt (a __syncthreads() would wait for the output stores)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  setup(tile, hf);
  int s0 = 0;
  stage_tile(0, 0, hf >= 0);
  stage_tile(1, 1, hf >= 0);
  wait_vm0();
  while (true) {
    const int tm0 = m0, tn0 = n0;
    int next, next_hf;
    const bool more = locate(it + 1, next, next_hf);
    const bool active = hf < 0 || wr == 0;   // half tile: the upper wave row has nothing to compute
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = s0 ^ (kt & 1);
      if constexpr (TUNE == 10) {          // ablation: NO synchronisation in the K loop (races by construction): what any re-ordering of
        if (kt == nk) block_sync();        // waits / barriers could gain at most; TUNE == 11: barrier only, TUNE == 12: DMA wait only
      } else if constexpr (TUNE == 11) {
        block_sync();
      } else if constexpr (TUNE == 12) {
        if (kt >= 2) wait_vm0();
      } else {
      if (kt >= 2) wait_vm0();  // own pieces of K-tile kt (issued one step ago); at kt == 2 also the previous tile's stores
      block_sync();             // K-tile kt visible to everyone; everyone is done with K-tile kt-1
      }
      // The buffer of K-tile kt-1 is free from here on: its 8 copies (K-tile kt+1, or K-tile 0 of the NEXT tile on the
      // last step) are issued BETWEEN this step's 32 MFMAs, and the two waves that share a SIMD (w, w+4) use alternating
      // slots -- a copy stalls its wave ~60-150 cycles at issue, which the partner's MFMAs cover; issued back to back by
      // all 8 waves right after the barrier they idle the whole CU for several hundred cycles per K-step.
      int ckt = kt + 1;
      bool do_copy = kt >= 1 && kt + 1 < nk;
      bool copy_half = hf >= 0;
      if (kt >= 1 && kt + 1 == nk && more) {  // last step: start the NEXT tile's first K-tile
        setup(next, next_hf);
        ckt = 0;
        do_copy = true;
        copy_half = next_hf >= 0;
      }
      if (!active) {  // idle wave row of a half tile: its share of the DMA, nothing else (the barriers above / below are taken by everybody)
        if (do_copy) {
#pragma unroll
          for (int c = 0; c < 8; ++c) copy_piece(c, ckt, cur ^ 1, copy_half);
        }
        continue;
      }
      const char* cA = smem + cur * 2 * TILE2_BYTES;
      const char* cW = cA + TILE2_BYTES;
      u32x4 fa[2][4], fb[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[0][j] = *(const u32x4*)(cW + lds_off(b_row[j], khalf));
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = *(const u32x4*)(cA + lds_off(a_row[i], khalf));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[(s + 1) & 1][j] = *(const u32x4*)(cW + lds_off(b_row[j], 2 * (s + 1) + khalf));
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[(s + 1) & 1][i] = *(const u32x4*)(cA + lds_off(a_row[i], 2 * (s + 1) + khalf));
        }
        // measurement build, TUNE 5 / 6: raise this wave's issue priority over its SIMD partner's for the 8 MFMAs of the sub-step (the partner
        // is then reading fragments or issuing copies), back to 0 for the fragment reads.  Measured neutral (round 3,
        // profiles/r3_gemm_setprio_experiment.txt): 971-998 TF/s against 986-1004 on the long shapes, bit-identical results
        if constexpr (TUNE == 5 || TUNE == 6) __builtin_amdgcn_s_setprio(TUNE == 5 ? 1 : 3);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            mma_chunk<T>(acc[i][j], fa[s & 1][i], fb[s & 1][j]);
            const int q = s * 8 + i * 2 + j;  // 0..31; see copy_slot
            if (do_copy) {
              if (copy_slot(TUNE, q, 0) >= 0 && pos == 0) copy_piece(copy_slot(TUNE, q, 0), ckt, cur ^ 1, copy_half);
              if (copy_slot(TUNE, q, 1) >= 0 && pos == 1) copy_piece(copy_slot(TUNE, q, 1), ckt, cur ^ 1, copy_half);
            }
          }
        if constexpr (TUNE == 5 || TUNE == 6) __builtin_amdgcn_s_setprio(0);
      }
    }
    block_sync();  // everyone is done with the last K-tile: its buffer takes the next tile's K-tile 1
    const int last = s0 ^ ((nk - 1) & 1);
    if (more) stage_tile(1, last, next_hf >= 0);
    s0 = last ^ 1;
    // epilogue; the two prefetched K-tiles must have landed before the first output store is issued (after that,
    // vmcnt also counts the stores and nobody waits on it until K-step 2 of the next tile)
    if (active) {
      const int mb = tm0 + wr * 128, nb = tn0 + wc * 64;
      float bias[4];
      load_bias4(g, nb + (lane & 15) * 4, bias);
      wait_vm0();
      // 8-row chunks through two alternating 2 KiB staging buffers per wave: the ds_writes of chunk c+1 are independent
      // of the ds_reads of chunk c, so LDS latency and the global stores of consecutive chunks overlap.  No hardware
      // wait is needed (DS operations of one wave execute in order); the compiler's order is pinned by wave_lds_order().
      auto stage_chunk = [&](float* st, const f32x16& a0, const f32x16& a1, int q) {
        wave_lds_order();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int row = r4 + 4 * (lane >> 5);
          st[row * 64 + (lane & 31)] = a0[4 * q + r4];
          st[row * 64 + 32 + (lane & 31)] = a1[4 * q + r4];
        }
        wave_lds_order();
      };
      auto run_epilogue = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const bool pf = FAST && MAP != ALPRO_MAP_FRAME_TOKENS && g.residual != nullptr;  // (FRAME_TOKENS: its row map + the ring would spill)  // residual rows are fetched one chunk ahead (see epi_prefetch_res)
        // Residual ring: RD - 1 chunks (2 KiB per wave each) are in flight ahead of the one being finished.
        constexpr int RD = 2;  // deeper (4: no change, 6: spills) -- profiles/r2_gemm_epilogue_experiments.txt item 5
        float4 ring[RD][2];
        if (pf) {
#pragma unroll
          for (int c0 = 0; c0 < RD - 1; ++c0) epi_prefetch_res<MAP>(g, mb + c0 * 8, nb, lane, ring[c0]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = i * 4 + q;
            if (pf && c + RD - 1 < 16) epi_prefetch_res<MAP>(g, mb + (c + RD - 1) * 8, nb, lane, ring[(c + RD - 1) % RD]);
            float* st = stage + (c & 1) * 512;
            stage_chunk(st, acc[i][0], acc[i][1], q);
            epi_rows16<T, ACT, MAP, FAST, 2>(g, st, mb + c * 8, nb, lane, bias, pf ? ring[c % RD] : nullptr);
          }
        }
      };
      const bool fast = epi_fast_ok(g, mb, 128, nb);
      bool c16 = false;
      if constexpr (sizeof(T) == 2 && MAP == ALPRO_MAP_IDENTITY)
        c16 = fast && g.c_dtype != ALPRO_F32 && ((g.ldc & 7) == 0) && (!g.C2 || (g.ldc2 & 7) == 0) && (!g.row_scale || g.row_scale_group >= 8);
      if (c16) {
        if constexpr (sizeof(T) == 2 && MAP == ALPRO_MAP_IDENTITY) {
          float bias8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) bias8[e] = g.bias ? g.bias[nb + (lane & 7) * 8 + e] : 0.f;
          // GELU_BWD: the saved pre-activation rows are fetched two chunks ahead of their use (same reason as the residual)
          u32x4 pring[3];
          auto load_pre = [&](int c) {
            return __builtin_nontemporal_load((const u32x4*)((const T*)g.C2 + (int64_t)(mb + c * 8 + (lane >> 3)) * g.ldc2 + nb + (lane & 7) * 8));
          };
          constexpr bool READS_C2 = ACT == ALPRO_ACT_GELU_BWD || ACT == ALPRO_ACT_MUL_SAVED;
          if (READS_C2) {
            pring[0] = load_pre(0);
            pring[1] = load_pre(1);
          }
          if constexpr (TUNE == 4) {  // ablation: no epilogue at all (accumulators kept live)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(acc[i][0]), "v"(acc[i][1]));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int c = i * 4 + q;
                if (READS_C2 && c + 2 < 16) pring[(c + 2) % 3] = load_pre(c + 2);
                float* st = stage + (c & 1) * 512;
                stage_chunk(st, acc[i][0], acc[i][1], q);
                epi_rows16_c16<T, ACT, 1, (TUNE == 3 ? 1 : 0)>(g, st, mb + c * 8, nb, lane, bias8, READS_C2 ? &pring[c % 3] : nullptr);
              }
            }
          }
        }
      } else if (fast) {
        run_epilogue(std::true_type{});
      } else {
        run_epilogue(std::false_type{});
      }
    }
    if (!more) break;
    tile = next;
    hf = next_hf;
    ++it;
    setup(tile, hf);  // recomputed (not kept live): frees the 16 source-pointer registers across the epilogue
  }
}


// ------------------------------------------------------------------------------------------------
// Round 4: the 256x256 tile as an 8-phase, two-group ("ping-pong") schedule on v_mfma_f32_16x16x32 fragments.
//
// What changes against gemm_nt256p_kernel (same macro tile, same 8 waves as 2 x 4, same 128x64 per wave, same LDS images and swizzle):
//  * the two waves that share a SIMD (w and w + 4) never want the matrix pipe at the same time.  A K-tile is four PHASES, one 64x32
//    quadrant of the wave's tile each (16 MFMAs on 8 independent accumulators: no dependent back-to-back issue); a phase is
//        LOAD segment: this quadrant's ds_read_b128 fragment reads + the two DMA pieces of one half-tile  -> s_barrier
//        MFMA segment: 16 MFMAs at raised priority                                                     -> s_barrier
//    and the upper wave row (waves 4-7) runs one barrier behind the lower one, so on every SIMD one wave is in its MFMA segment while
//    its partner issues LDS reads and copies -- the pipe sees a continuous MFMA stream instead of two identical streams colliding;
//  * the stage buffers are eight 16 KiB half-tile slots (2 K-tile parities x {A rows 0-127, A 128-255, W 0-127, W 128-255}) refilled one
//    slot per phase as soon as its last reader is two phases behind; ONE counted wait per K-tile (vmcnt(2) at the end of phase 4's LOAD
//    segment: everything but the half-tile just issued has landed), placed one phase before the first read of the data it covers;
//  * the next tile's first K-tile streams in during the current tile's last K-tile, so the K loop runs across tile boundaries without a
//    prologue; the two groups re-align only for the epilogue (both store at the same time) and split again behind it.
// Schedule (K-tile t of the tile, buffer parity P = t & 1; quadrant = (A rows mi*64.., W rows ni*32..) of the wave's 128 x 64):
//    phase 1  reads B(ni 0) + A(mi 0)   copies A-half 0 of K-tile t+1 -> parity P^1     MFMA quadrant (0, 0)
//    phase 2  reads B(ni 1)             copies A-half 1 of K-tile t+1 -> parity P^1     MFMA quadrant (0, 1)
//    phase 3  reads A(mi 1)             copies W-half 1 of K-tile t+1 -> parity P^1     MFMA quadrant (1, 1)
//    phase 4  --                        copies W-half 0 of K-tile t+2 -> parity P, vmcnt(2)   MFMA quadrant (1, 0)
// Hazards (phase index k, barrier b; lower group: LOAD(k) in [b 2k-1, b 2k], MFMA(k) in [2k, 2k+1]; upper group one barrier later):
//    RAW  a wave's wait at the end of LOAD(k) precedes barrier 2k+1 for both groups; the data is first read in LOAD(k+1), after it;
//    WAR  a slot last read in LOAD(kr) is idle once barrier 2kr+2 has passed (the upper group's reads retire inside its MFMA(kr));
//         its refill is issued in LOAD(kw), kw >= kr + 2, i.e. after barrier 2kw-1 >= 2kr+3.  (A-half h: kr = phase 3 of K-tile t-1, kw =
//         phase 1 / 2 of K-tile t; W-half 1: kr = phase 2 of t-1, kw = phase 3 of t; W-half 0: kr = phase 2 of t, kw = phase 4 of t.)
// Would launch_gemm send this descriptor to the 8-phase kernel (gemm_nt256q_kernel)?  Eligible: 16-bit operands, identity map, 16-bit output
// through the 16-byte-store epilogue or fp32 output through the fp32 one, whole 256-column tiles, 128-byte-aligned operand rows, an even number
// >= 4 of 64-deep K-tiles, 32-bit operand offsets, >= 160 whole tiles (below that the 128 x 128 kernel fills the chip better).
// Ragged M: when the remainder is a multiple of 16 rows the kernel takes the partial tile row itself (invalid copy pieces re-read valid rows,
// the epilogue skips rows beyond M) -- ONE launch; the round-4 first form ran the remainder as a second launch on the 128 x 128 kernel: 46
// launches of ~46 us per training step at B = 64 (M = 100416 = 392 * 256 + 64), serialised behind the main kernel.  Other remainders still
// take that split (the remainder launch carries m_off: row scale / dropout index by absolute row).
static bool q_kernel_takes(const alpro_gemm_desc_t& g, bool* ragged_in_kernel) {
  constexpr int BM2q = 256, BN2q = 256;
  if (g.dtype == ALPRO_F32 || g.map_mode != ALPRO_MAP_IDENTITY) return false;
  const int m_full = g.M / BM2q * BM2q, m_rem = g.M - m_full;
  const bool c16 = g.c_dtype != ALPRO_F32 && (g.ldc & 7) == 0 && (!g.C2 || g.c2_tiled || (g.ldc2 & 7) == 0) && (!g.residual || (g.ldr & 3) == 0) && ((uintptr_t)g.C % 16) == 0 &&
                   (!g.row_scale || g.row_scale_group >= 8);   // (a pass of 8 rows takes its scales from at most two groups: epi_rows16_c16)
  const bool c32 = g.c_dtype == ALPRO_F32 && g.act == ALPRO_ACT_NONE && !g.C2 && !g.drop_seed && (g.ldc & 3) == 0 && (!g.residual || ((g.ldr & 3) == 0 && ((uintptr_t)g.residual % 16) == 0)) &&
                   ((uintptr_t)g.C % 16) == 0 && (!g.row_scale || g.row_scale_group >= 16);   // (the fp32 epilogue takes a fragment row's scales from at most two groups)
  const bool shape = (g.N % BN2q) == 0 && ((g.lda * 2) % 128) == 0 && ((g.ldw * 2) % 128) == 0 && (g.K % 128) == 0 && g.K >= 256 &&
                     (int64_t)g.M * g.lda * 2 < (int64_t)0xFFFF0000 && (int64_t)g.N * g.ldw * 2 < (int64_t)0xFFFF0000;
  *ragged_in_kernel = m_rem > 0 && (m_rem % 16) == 0;
  const int full_tiles = (g.N / BN2q) * (m_full / BM2q);
  const int force = get_option(OPT_GEMM_TILE);
  return get_option(OPT_GEMM_KIND) == 1 && (c16 || c32) && shape && (force ? force == 256 : full_tiles >= 160);
}
// ... and may its C2 buffer be in the tile layout (alpro_hip.h: c2_tiled)?  Every tile of the output must go through the kernel's packed
// epilogue: the GELU_SAVE_GRAD / MUL_SAVED pair without anything that epilogue does not do, and no second launch for a remainder.
static bool c2_tiled_ok(const alpro_gemm_desc_t& g) {
  const int m_rem = g.M % 256;
  return (g.act == ALPRO_ACT_GELU_SAVE_GRAD || g.act == ALPRO_ACT_MUL_SAVED) && g.C2 && ((uintptr_t)g.C2 % 16) == 0 && g.c_dtype != ALPRO_F32 && !g.residual && !g.row_scale &&
         !g.drop_seed && (m_rem % 16) == 0;
}

constexpr int HALF2_BYTES = 128 * ROWB;           // 16 KiB: 128 rows x 128 bytes
constexpr int STAGE2_BYTES = 4 * HALF2_BYTES;     // one K-tile parity: A0 | A1 | W0 | W1

template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
  static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma16<f16_t> {
  static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

template <typename T, int ACT, int MAP>
__global__ __launch_bounds__(NT2, 2) __attribute__((amdgpu_num_vgpr(127))) void gemm_nt256q_kernel(const alpro_gemm_desc_t g, const TileSched sc) {
  static_assert(sizeof(T) == 2, "16-bit operands only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (g.N + BN2 - 1) / BN2, ntm = (g.M + BM2 - 1) / BM2;
  const int nblk = ntn * ntm;
  const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  const int nk = g.K >> 6;                       // K-tiles of 64 elements; even and >= 4 (launcher)
  const uint32_t lds_base = lds_addr_of(smem);

  // ---- which tiles (round 5) ----------------------------------------------------------------------------------------------------
  // Tiles are numbered row-panel-major (tile = tm * ntn + tn) and dealt to the 8 XCDs in chunks of 32: XCD y's LIST is
  //     j -> tile (j / 32) * 256 + y * 32 + (j % 32),        j = 0, 1, 2, ...  while that is < nblk,
  // i.e. the tiles the round-4 static walk (workgroup slot s of 256 takes s, s + 256, ...) gave to the XCD's 32 workgroups, in the order
  // it visited them: at any moment an XCD works on a contiguous run of tiles, so the tn tiles of an A row panel meet in ONE L2.
  //   gemm_sched 0 (sc.blk == nullptr): workgroup idx of the XCD takes j = idx, idx + p, idx + 2p, ... (p = gridDim.x / 8) -- with 256
  //     workgroups exactly that static walk.
  //   gemm_sched 1: the first TWO tiles of a workgroup are the static ones (j = idx, idx + p: no atomic stands between the launch and the first
  //     MFMA); every further j comes from the XCD's ticket counter, j = 2p + ticket: one agent-scope atomic per tile, issued by wave 0 behind
  //     an epilogue two tiles ahead of the tile it pays for and read back behind the next K loop -- nothing is added to the K loop.
  //     A workgroup that cannot be resident -- another kernel holds its CU: RCCL's channels during the overlapped gradient exchange, a side
  //     stream -- draws no tickets: the resident ones finish its share of the lists one tile at a time instead of the launch waiting a whole
  //     extra round for it (profiles/r4_overlap_cu_contention.txt: +42 % with 8 of 256 CUs taken).  Its two STATIC tiles are covered by a
  //     claim word per workgroup: a workgroup claims its own pair with one atomic at its start (the answer is awaited by the pipeline fill's
  //     own wait), and a workgroup that has run out of work -- own list dry: it then looks at all eight counters and all claim words with
  //     ONE pair of loads -- takes tickets of other XCDs' lists and, after those, the pair of a workgroup that has not started yet.
  //     Nobody waits for anybody; the block of counters / claim words is zeroed by the NEXT launch of the same stream (TileSched::prev).
  // Results do not depend on who computes a tile: bitwise identical under either walk.
  const uint32_t p = gridDim.x >> 3;
  const uint64_t t_start = wall_clock64();
  constexpr uint32_t RESCUE_TICKS = 1000;   // 10 us of the 100 MHz wall clock
  auto list_tile = [&](int y, uint32_t j) -> int {
    const uint32_t t = ((j >> 5) << 8) + ((uint32_t)y << 5) + (j & 31u);
    return (j < 0x100000u && t < (uint32_t)nblk) ? (int)t : -1;
  };
  auto list_len = [&](int y) -> int {   // number of valid positions of XCD y's list
    const int rem = (nblk & 255) - 32 * y;
    return (nblk >> 8) * 32 + (rem < 0 ? 0 : (rem > 32 ? 32 : rem));
  };
  // Mailbox wave 0 -> everybody: two dwords at the start of the A-half-1 slot of parity 1.  That slot's last reader is phase 3 of a tile's
  // last K-tile and its next writer the copy of phase 2 of the following tile's first K-tile (two barriers into that tile): dead in between.
  // (an LDS-address-space pointer: through a generic pointer the accesses become FLAT instructions, which count on vmcnt AND lgkmcnt and made
  // every read wait for the epilogue's stores)
  typedef __attribute__((address_space(3))) volatile int lds_int_t;
  lds_int_t* mbox = (lds_int_t*)(__attribute__((address_space(3))) char*)(smem + STAGE2_BYTES + HALF2_BYTES);
  // walk state (wave 0's copy is the one that counts).  Dynamic: how many XCD lists have run dry for this workgroup (tickets are drawn from
  // XCD (own + wstate) % 8).  Static: the workgroup's next list position.
  uint32_t wstate = sc.blk ? 0u : (uint32_t)(blockIdx.x >> 3) + 2 * p;
  // Tickets a workgroup may still draw AHEAD (pipelined, two tiles before it can start them): its fair share of its own list,
  // ceil((len - 2p) / p).  Without the cap a workgroup that runs a few hundred ns ahead of a neighbour draws the list's last ticket while the
  // neighbour still has two tiles to go, and the launch ends one tile later than the static walk (measured on the qkv shape at B = 64, whose
  // lists divide exactly: +8 %).  Whatever is left when a workgroup is OUT of work -- tickets of workgroups that never started, the other
  // XCDs' lists -- goes through steal(), one tile at a time, to whoever is idle then.
  int quota = 0;
  bool pending = false;   // a pipelined ticket is in flight
  if (sc.blk) {
    const int mine = list_len((int)(blockIdx.x & 7u)) - (int)(2 * p);
    quota = mine > 0 ? (mine + (int)p - 1) / (int)p : 0;
  }
  // One returning atomic, lane 0 of wave 0 only (`on`; otherwise the instruction runs with an empty EXEC mask): `add` = a ticket of the
  // current list's counter, else the claim (atomic or) of workgroup `w`'s word.  The value lands in `r` when the memory system answers:
  // whoever reads it waits first (s_waitcnt vmcnt), like for the copies.  These two forms are for the BLOCKING draws of steal(): `r` is read
  // behind a vmcnt(0) a few instructions on ("+v": one register from the atomic to its reader; a CPU test checks the built ISA).
  auto ticket_issue = [&](uint32_t& r, bool on) {
    uint64_t sv;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(on ? 1 : 0);
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %5\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                 : "+v"(r), "=&s"(sv) : "v"(((blockIdx.x + wstate) & 7u) * 4u), "v"(1u), "s"(sc.blk), "s"(m) : "memory");
  };
  auto claim_issue = [&](uint32_t& r, uint32_t w, uint32_t bits, bool on) {   // bit 0 / bit 1: the first / second static tile of workgroup w
    uint64_t sv;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(on ? 1 : 0);
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %5\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_or %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                 : "+v"(r), "=&s"(sv) : "v"((SCHED_CLAIM0 + w) * 4u), "v"(bits), "s"(sc.blk), "s"(m) : "memory");
  };
  // The PIPELINED draws (the static pair's claim, the ticket for the tile after next) answer into v255, a register the compiler does not own
  // (the kernel is built with amdgpu_num_vgpr(127): on gfx90a+ the number counts per register-file half, i.e. v0-v253 are the compiler's): their answers are in flight across a pipeline fill / a whole K loop, and a compiler-owned register
  // may be copied or re-assigned at any block boundary in between -- a copy of a register with an atomic in flight copies the OLD contents
  // (it happened whenever an epilogue variant was added: the allocator split the live range).  tk_read() is placed behind a counted wait
  // that was issued after the atomic.
  auto ticket_issue_tk = [&](bool on) {
    uint64_t sv;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(on ? 1 : 0);
    // reserved-register site (deliberate; -Werror=inline-asm otherwise): v255 is outside the compiler's budget (amdgpu_num_vgpr(127)) and holds the atomic's answer while it is in flight; tests/test_host_cpu.py checks the built ISA (nothing else names v254 / v255, every instantiation is allocated 256 registers)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add v255, %1, %2, %3 sc0\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "v"(((blockIdx.x + wstate) & 7u) * 4u), "v"(1u), "s"(sc.blk), "s"(m) : "memory", "v255");   // (the clobber is what makes the compiler COUNT v255 into the kernel's register allocation -- without it an instantiation that needs 240 registers gets 240 and the atomic writes outside the wave's file; the "reserved register" warning is expected)
#pragma clang diagnostic pop
  };
  auto claim_issue_tk = [&](uint32_t w, uint32_t bits, bool on) {
    uint64_t sv;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(on ? 1 : 0);
    // reserved-register site (deliberate; -Werror=inline-asm otherwise): v255 is outside the compiler's budget (amdgpu_num_vgpr(127)) and holds the atomic's answer while it is in flight; tests/test_host_cpu.py checks the built ISA (nothing else names v254 / v255, every instantiation is allocated 256 registers)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_or v255, %1, %2, %3 sc0\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "v"((SCHED_CLAIM0 + w) * 4u), "v"(bits), "s"(sc.blk), "s"(m) : "memory", "v255");   // (the clobber is what makes the compiler COUNT v255 into the kernel's register allocation -- without it an instantiation that needs 240 registers gets 240 and the atomic writes outside the wave's file; the "reserved register" warning is expected)
#pragma clang diagnostic pop
  };
  auto tk_read = [&]() -> uint32_t {   // lane 0's answer (wave 0)
    uint32_t r;
    asm volatile("v_readfirstlane_b32 %0, v255" : "=s"(r) : : "memory");
    return r;
  };
  // ticket -> tile of the list tickets are currently drawn from; a dry list moves the workgroup on to the next XCD's
  auto ticket_tile = [&](uint32_t k) -> int {
    int t;
    if (sc.blk) {
      t = list_tile((int)((blockIdx.x + wstate) & 7u), 2 * p + k);
      if (t < 0) ++wstate;
    } else {
      t = list_tile((int)(blockIdx.x & 7u), wstate);
      wstate += p;
    }
    return t;
  };
  // A workgroup out of work (dynamic walk; every wave calls it, one barrier; no copy in flight: the caller drained vmcnt).  Wave 0 reads the
  // eight counters and the claim words (five loads in flight together), then
  //   * takes ONE ticket of the first list (own XCD's first) that still has positions left -- one, not a pair: the last partial round then
  //     spreads over everybody who is out of work instead of the first arrivals taking two tiles each --, or, when every list is dry,
  //   * claims ONE static tile nobody has claimed yet (a workgroup that could not start: some other kernel holds its CU).  Which one is drawn
  //     from a hash of the workgroup id over all open tiles, so that a few hundred helpers arriving together do not all go for the same word.
  //     Rescue waits until RESCUE_TICKS after this workgroup's own start (`t_start`, 100 MHz wall clock): by then every workgroup that CAN be
  //     resident has started and claimed its pair (a later rescue of a workgroup that starts at that very moment is still correct: the claim
  //     atomic arbitrates; it only costs that workgroup its pipeline fill),
  // and posts the tile (-1: nothing left anywhere).  A ticket that comes back beyond its list, or a claim somebody else won, means the picture
  // was stale: look again.
  auto steal = [&](int& t0, int& t1) {
    if (wave == 0) {
      int a = -1;
      for (int tries = 0; tries < 96 && a < 0; ++tries) {
        const uint32_t cnt = lane < 8 ? __hip_atomic_load(sc.blk + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        u32x4 clm;   // claim words of workgroups 4 lane .. 4 lane + 3 (agent-scope loads: the words are set by other XCDs' atomics)
        clm.x = __hip_atomic_load(sc.blk + SCHED_CLAIM0 + 4 * lane + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        clm.y = __hip_atomic_load(sc.blk + SCHED_CLAIM0 + 4 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        clm.z = __hip_atomic_load(sc.blk + SCHED_CLAIM0 + 4 * lane + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        clm.w = __hip_atomic_load(sc.blk + SCHED_CLAIM0 + 4 * lane + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int y = -1;
        for (int i = 0; i < 8 && y < 0; ++i) {
          const int c = (int)((blockIdx.x + i) & 7u);
          if ((int)__builtin_amdgcn_readlane(cnt, c) < list_len(c) - (int)(2 * p)) y = c;
        }
        if (y >= 0) {
          uint32_t k0 = 0;
          wstate = (uint32_t)((y - (int)(blockIdx.x & 7u)) & 7);   // tickets are drawn from (own + wstate) % 8 from here on
          ticket_issue(k0, true);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          a = list_tile(y, 2 * p + __builtin_amdgcn_readfirstlane(k0));
          continue;
        }
        // open static tiles: slot s = 2 i + b of a lane is tile b (0 = first, 1 = second) of workgroup 4 lane + i; workgroups below gridDim.x count
        const uint32_t w0 = 4u * lane;
        const uint32_t words[4] = {clm.x, clm.y, clm.z, clm.w};
        uint64_t open[8];
        int total = 0;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
          open[sl] = __builtin_amdgcn_ballot_w64(w0 + (sl >> 1) < gridDim.x && ((words[sl >> 1] >> (sl & 1)) & 1u) == 0u);
          total += __builtin_popcountll(open[sl]);
        }
        if (total == 0) break;
        while ((uint32_t)(wall_clock64() - t_start) < RESCUE_TICKS) __builtin_amdgcn_s_sleep(8);
        int q = (int)(((blockIdx.x + 1u) * 0x9E3779B1u >> 12) % (uint32_t)total);   // the q-th open tile, q spread over the helpers
        uint32_t w = 0, bit = 0;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
          const int n = __builtin_popcountll(open[sl]);
          if (bit == 0 && q < n) {
            uint64_t msk = open[sl];
            for (int i = 0; i < q; ++i) msk &= msk - 1;
            w = 4u * (uint32_t)__builtin_ctzll(msk) + (uint32_t)(sl >> 1);
            bit = 1u << (sl & 1);
          }
          q -= n;
        }
        uint32_t old = 3;
        claim_issue(old, w, bit, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((__builtin_amdgcn_readfirstlane(old) & bit) == 0) a = list_tile((int)(w & 7u), (w >> 3) + (bit == 2u ? p : 0u));   // (-1: that workgroup had no such tile -- look again)
      }
      const int b = -1;
      mbox[0] = a;
      mbox[1] = b;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    t0 = __builtin_amdgcn_readfirstlane(mbox[0]);
    t1 = __builtin_amdgcn_readfirstlane(mbox[1]);
  };

  // DMA sources.  Nothing is clamped per lane and everything tile-dependent is wave-uniform: a copy reads
  //   [A + (m0 + h*128) * lda_b + kt*128]  (SGPR pair)  +  [((r0 + i*8) * lda_b + chunk*16) ^ i*64]  (one 32-bit VGPR per piece)
  // for piece i of half-tile h, r0 = wave*16 + (lane >> 3) = the lane's row in piece 0, chunk = (lane & 7) ^ swizzle(r0); piece 1 sits 8 rows
  // further, where the swizzle differs by 4 chunks = 64 bytes (lda_b is a multiple of 128: launcher).
  const int r0 = wave * 16 + (lane >> 3);
  const uint32_t sw0 = (uint32_t)(((lane & 7) ^ ((r0 >> 1) & 7)) << 4);
  uint32_t oa[2], ow[2][2];   // A: [piece i] (the half-tile's 128 rows sit in its base pointer); W: [half-tile h][piece i]: the lane's byte offset
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    oa[i] = (uint32_t)((r0 + i * 8) * lda_b) + (sw0 ^ (uint32_t)(i * 64));
#pragma unroll
    for (int h = 0; h < 2; ++h) ow[h][i] = (uint32_t)((r0 + h * 128 + i * 8) * ldw_b) + (sw0 ^ (uint32_t)(i * 64));
  }
  // Tile bases (wave-uniform).  a[h] = where THIS WAVE's 16 rows of A half-tile h start, minus the wave's own row offset (which oa carries):
  // normally A + (m0 + h * 128) * lda_b.  Ragged last tile row (M % 256 = vr valid rows, a multiple of 16: launcher): a wave's two copy
  // instructions per half-tile move rows [h * 128 + wave * 16, + 16) -- valid or not as a whole -- and an invalid group re-reads rows 0-15 of
  // the tile instead (base moved back by wave * 16 rows): finite values that only reach accumulator rows the epilogue never stores.  All of it
  // is folded into the per-tile base: the copies in the K loop cost what they cost on a full tile.
  struct Tile { const char* a[2]; const char* w; };
  auto tile_base = [&](int tile) {
    const int tm = ntn == 1 ? tile : (int)__umulhi((uint32_t)tile, sc.magic_ntn), tn = tile - tm * ntn;
    const int vr = g.M - tm * BM2;   // (>= 256 on full tiles)
    const char* a0 = (const char*)g.A + (int64_t)tm * BM2 * lda_b;
    Tile t;
#pragma unroll
    for (int h = 0; h < 2; ++h) t.a[h] = a0 + (int64_t)((h * 128 + wave * 16 < vr) ? h * 128 : -(wave * 16)) * lda_b;
    t.w = (const char*)g.W + (int64_t)tn * BN2 * ldw_b;
    return t;
  };
  // half-tile `hs` (0 / 1: A rows 0-127 / 128-255, 2 / 3: W rows) from `kbase` (= that half's tile base + kt * 128 bytes, wave-uniform) -> slot hs of parity `par`
  auto copy_half = [&](const char* kbase, int hs, int par) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t vo = hs < 2 ? oa[i] : ow[hs & 1][i];
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + par * STAGE2_BYTES + hs * HALF2_BYTES + (wave * 2 + i) * 1024);
      // reserved-register site (the product is built with -Werror=inline-asm; this one is deliberate): global_load_lds takes its LDS address from m0; listing it as clobbered is what keeps the compiler from assuming a value of its own survives the statement (it writes m0 itself before each of its own uses: LDS-DMA builtins, s_movrel); the K-loop ISA tests of tests/test_host_cpu.py read the built object
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(kbase), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  // fragment read offsets: lane l supplies row (l & 15) and the 8-element k group (l >> 4) of a 16 x 32 operand fragment; chunk index
  // (ks * 4 + kg) ^ swizzle(row) == (ks * 64 bytes) ^ ((kg ^ swizzle) * 16 bytes)
  const int l15 = lane & 15, kg = lane >> 4;
  const int frag0 = l15 * ROWB + ((kg ^ ((l15 >> 1) & 7)) << 4);
  const char* aF[2] = {smem + wr * HALF2_BYTES + frag0, smem + wr * HALF2_BYTES + (frag0 ^ 64)};
  const char* bF[2] = {smem + (2 + (wc >> 1)) * HALF2_BYTES + (wc & 1) * 64 * ROWB + frag0,
                       smem + (2 + (wc >> 1)) * HALF2_BYTES + (wc & 1) * 64 * ROWB + (frag0 ^ 64)};
  float* stage = (float*)(smem + 2 * STAGE2_BYTES + wave * (16 * 64 * 4));   // 4 KiB per wave: 16 rows x 64 columns fp32

  auto barrier = [] {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // (the ticket in flight -- lane 0 of wave 0, v255: drawn behind tile i - 1's epilogue for tile i + 2, read back behind tile i's K loop)
  // the static pair, and (dynamic walk) its claim: in flight under the pipeline fill.  Workgroup 0 also hands the block of this stream's
  // PREVIOUS launch back zeroed (that launch is complete: same stream).
  int cur_t = list_tile((int)(blockIdx.x & 7u), blockIdx.x >> 3), nxt_t = list_tile((int)(blockIdx.x & 7u), (blockIdx.x >> 3) + p);
  if (sc.prev && blockIdx.x == 0 && wave == 0) {
#pragma unroll
    for (int i = 0; i < (SCHED_BLOCK_U32 + 63) / 64; ++i)
      if (i * 64 + lane < SCHED_BLOCK_U32) sc.prev[i * 64 + lane] = 0u;
  }
  claim_issue_tk(blockIdx.x, 3u, wave == 0 && sc.blk);   // (the answer travels in the ticket register: the first ticket is drawn after it has been read)
  bool fresh = sc.blk != nullptr;   // the static pair's claim is in flight
  bool have = cur_t >= 0;
  while (true) {
  if (!have) {
    if (!sc.blk) break;
    fresh = false;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (steal()'s own waits count on an empty queue)
    steal(cur_t, nxt_t);
    if (cur_t < 0) break;
    quota = 0;   // from here on one tile at a time, when idle
  }
  have = false;
  Tile cur = tile_base(cur_t);
  Tile nxt = tile_base(nxt_t >= 0 ? nxt_t : cur_t);   // (no next tile: the run-ahead copies re-read this tile's first K-tiles into dead slots)
  // pipeline fill: K-tile 0 complete in parity 0, W-half 0 of K-tile 1 on its way into parity 1
#pragma unroll
  for (int hs = 0; hs < 4; ++hs) copy_half(hs < 2 ? cur.a[hs] : cur.w, hs, 0);
  copy_half(cur.w + ROWB, 2, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (fresh && wave == 0) {   // are the static tiles still this workgroup's?  (wave 0's wait above covered the claim)
    mbox[0] = (int)(tk_read() & 3u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  barrier();
  if (fresh) {
    fresh = false;
    const int gone = __builtin_amdgcn_readfirstlane(mbox[0]);   // bit 0 / 1: somebody rescued the first / second one while this workgroup waited for a CU
    if (gone) {
      barrier();   // (everybody has read the mailbox before it is written again)
      int t0 = (gone & 1) ? -1 : cur_t, t1 = (gone & 2) ? -1 : nxt_t;
      if (t0 < 0) { t0 = t1; t1 = -1; }
      cur_t = t0;
      nxt_t = t1;
      have = cur_t >= 0;
      continue;   // fill the pipeline again for what is left, or look for other work
    }
  }
  pending = sc.blk && nxt_t >= 0 && quota > 0;
  quota -= pending ? 1 : 0;
  ticket_issue_tk(wave == 0 && pending);   // for the tile after next

  while (true) {
    const int tile = cur_t;
    const int tm0 = (ntn == 1 ? tile : (int)__umulhi((uint32_t)tile, sc.magic_ntn)) * BM2, tn0 = (tile - (tm0 >> 8) * ntn) * BN2;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wr == 1) barrier();   // the upper wave row drops one barrier behind

    // one K-tile (parity P compile-time): four phases
    auto ktile = [&](auto par_tag, int t) {
      constexpr int P = decltype(par_tag)::value;
      u32x4 fa[4][2], fb[2][2][2];
      const bool in1 = t + 1 < nk, in2 = t + 2 < nk;          // targets inside this tile? else the next tile's K-tile 0 / 1
      const int k1 = in1 ? t + 1 : 0, k2 = in2 ? t + 2 : t + 2 - nk;
      const char* a10 = (in1 ? cur.a[0] : nxt.a[0]) + (int64_t)k1 * ROWB;   // K-tile t+1: both A halves and W half 1
      const char* a11 = (in1 ? cur.a[1] : nxt.a[1]) + (int64_t)k1 * ROWB;
      const char* w1 = (in1 ? cur.w : nxt.w) + (int64_t)k1 * ROWB;
      const char* w2 = (in2 ? cur.w : nxt.w) + (int64_t)k2 * ROWB;   // K-tile t+2: W half 0
      auto load_a = [&](int mi) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) fa[f][ks] = *(const u32x4*)(aF[ks] + P * STAGE2_BYTES + (mi * 64 + f * 16) * ROWB);
      };
      auto load_b = [&](int ni) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) fb[ni][f][ks] = *(const u32x4*)(bF[ks] + P * STAGE2_BYTES + (ni * 32 + f * 16) * ROWB);
      };
      auto mma = [&](int mi, int ni) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[mi * 4 + f][ni * 2 + e] = Mma16<T>::run(fa[f][ks], fb[ni][e][ks], acc[mi * 4 + f][ni * 2 + e]);
        __builtin_amdgcn_s_setprio(0);
      };
      // phase 1
      load_b(0);
      load_a(0);
      copy_half(a10, 0, P ^ 1);
      barrier();
      mma(0, 0);
      barrier();
      // phase 2
      load_b(1);
      copy_half(a11, 1, P ^ 1);
      barrier();
      mma(0, 1);
      barrier();
      // phase 3
      load_a(1);
      copy_half(w1, 3, P ^ 1);
      barrier();
      mma(1, 1);
      barrier();
      // phase 4
      copy_half(w2, 2, P);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      barrier();
      mma(1, 0);
      barrier();
    };
    for (int t = 0; t < nk; t += 2) {
      ktile(std::integral_constant<int, 0>{}, t);
      ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    // Where the NEXT ticket is drawn.  Behind the epilogue it has ~2 us until the next K loop's first counted wait, which is in-order: it also
    // waits for this atomic, and 32 workgroups of an XCD that run in lockstep hit their counter together.  An epilogue without loads has no
    // wait of its own behind its first fragment row (the bias values are the only thing it fetches), so there the ticket goes out right
    // behind that row (another ~3 us of slack); the epilogues that fetch rows all along (saved factor, fp32 residual: their counted waits
    // would stall on the atomic) keep drawing behind themselves.  (Measured effect of the early draw: within noise.  Of the +6 % of the B = 64
    // qkv shape it was introduced against, half was the first-measurement-of-the-process artefact of the probe -- clocks still settling,
    // profiles/r5_gemm_stagger_probe.txt -- and +3 % is still there on that shape with a warm-up: profiles/r5_gemm_sched_contention.txt.)
    constexpr bool EPI_LOADS = ACT == ALPRO_ACT_GELU_BWD || ACT == ALPRO_ACT_MUL_SAVED;
    constexpr bool PK_ACT = MAP == ALPRO_MAP_IDENTITY && (ACT == ALPRO_ACT_NONE || ACT == ALPRO_ACT_GELU || ACT == ALPRO_ACT_RELU || ACT == ALPRO_ACT_GELU_SAVE_GRAD ||
                                                        ACT == ALPRO_ACT_MUL_SAVED);
    // (the saved-factor multiply takes it only when the factor lies in the tile layout -- c2_tiled, below; the launcher refuses a tiled
    // descriptor this test would send down the staged path)
    const bool pk = PK_ACT && (sc.epi != 0 || g.c2_tiled) && g.c_dtype != ALPRO_F32 && !g.residual && !g.row_scale && !g.drop_seed &&
                    (ACT == ALPRO_ACT_GELU_SAVE_GRAD || (ACT == ALPRO_ACT_MUL_SAVED ? g.c2_tiled != 0 : !g.C2));
    const bool early = !EPI_LOADS && !g.residual;
    int nn_w0 = -1;
    if (wave == 0) {   // the tile after next: the ticket drawn a tile ago has landed (every counted wait of this K loop was issued behind it)
      nn_w0 = (pending || (!sc.blk && nxt_t >= 0)) ? ticket_tile(tk_read()) : -1;   // (no ticket drawn / a dry list: -1 -- steal() behind the next tile looks further)
      mbox[0] = nn_w0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    auto early_ticket = [&] {   // wave 0, behind the first fragment row of a load-free epilogue
      if (wave == 0 && early) {
        pending = sc.blk && nn_w0 >= 0 && nxt_t >= 0 && quota > 0;
        quota -= pending ? 1 : 0;
        ticket_issue_tk(pending);
      }
    };
    if (wr == 0) barrier();   // re-align: both wave rows run their epilogues at the same time

    // ---- epilogue: one 16-row fragment row (16 x 64 fp32 = 4 KiB of wave-private LDS) at a time
    {
      // The epilogue's own copy of the lane id, opaque to the optimiser: everything lane-derived below (staging offsets, row / column pieces,
      // output pointers) is then computed HERE, where the 64 fragment registers are free -- hoisted over the K loop as tile-loop invariants
      // they were spilled at kernel start and reloaded per fragment row (a scratch load returns behind every store issued before it).
      int le = lane;
      asm volatile("" : "+v"(le));
      const int l15 = le & 15, kg = le >> 4;
      const int mb = tm0 + wr * 128, nb = tn0 + wc * 64;
      auto rows_ok = [&](int mf) { return mb + mf * 16 < g.M; };   // (wave-uniform)
      auto stage_rows = [&](int mf) {
        wave_lds_order();   // the other lanes' reads of the previous fragment row are issued before these writes ...
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) stage[(kg * 4 + r) * 64 + nf * 16 + l15] = acc[mf][nf][r];
        wave_lds_order();   // ... and every write of this one before the reads that follow
      };
      if (g.c_dtype != ALPRO_F32) {
        if constexpr (MAP == ALPRO_MAP_IDENTITY) {
          constexpr bool READS_C2 = ACT == ALPRO_ACT_GELU_BWD || ACT == ALPRO_ACT_MUL_SAVED;
          // The PACKED path (round 5): bias / activation / conversion to 16 bits happen in the accumulator layout, and what crosses the LDS is
          // the 16-bit result -- two fragment rows (32 x 64) per pass as 8-byte units of four rows x one column, one ds_write_b64 per
          // fragment instead of four ds_write_b32 (the staging writes are what an epilogue costs first: 128 ds_write_b32 per wave at 4 LDS
          // cycles each = 2 us per tile, all eight waves on the one LDS pipe), four ds_read_b128 per lane (8 columns x 4 rows) and 16 v_perm
          // to turn them into four 16-byte row pieces.  For the epilogues that need nothing in the OUTPUT layout: no residual, row scale,
          // dropout or saved factor, full fragment rows.  The 16-byte pieces of a 64-byte block are XORed with the row group so that the 16
          // lanes of a ds_read_b128 group hit 16 different slots.  Same values as the staged fp32 path (every step is elementwise).
          if (pk && mb >= g.M) {
            // (a wave of the last tile row without a valid row)
          } else if (pk) {
            if constexpr (PK_ACT) {
              typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
              typedef __attribute__((address_space(3))) char lds_char_t;
              lds_char_t* st8 = (lds_char_t*)(smem + 2 * STAGE2_BYTES) + wave * (16 * 64 * 4);
              float bcol[4];
#pragma unroll
              for (int nf = 0; nf < 4; ++nf) bcol[nf] = g.bias ? g.bias[nb + nf * 16 + l15] : 0.f;
              const int kgx = le >> 3, cg = le & 7;
              const int woff = kg * 512 + (l15 >> 3) * 64 + ((((l15 >> 1) & 3) ^ kg) << 4) + (l15 & 1) * 8;   // + f * 2048 + nf * 128
              const int roff = kgx * 512 + cg * 64;                                                            // + ((j ^ (kgx & 3)) << 4)
              const int64_t ldc = g.ldc, ldc2 = g.ldc2;
              T* Cb = (T*)g.C + (int64_t)(mb + 4 * kgx) * ldc + nb + 8 * cg;
              T* C2b = ACT == ALPRO_ACT_GELU_SAVE_GRAD ? (T*)g.C2 + (int64_t)(mb + 4 * kgx) * ldc2 + nb + 8 * cg : nullptr;
              // The saved factor in the TILE layout (c2_tiled; gelu' of fc1, written by the GELU_SAVE_GRAD form and read back by the MUL_SAVED
              // dgrad of fc2 -- nobody else looks at it): element (fragment row mf, fragment nf, row r of the lane's four, lane) of wave w of
              // tile t lives at ((t * 8 + w) * 8 + mf) * 1024 + (nf >> 1) * 512 + lane * 8 + (nf & 1) * 4 + r -- i.e. exactly the accumulator
              // registers, 16 bits each, two 16-byte pieces per lane and fragment row, 1 KiB contiguous per store / load instruction.  Neither
              // kernel sends it through the LDS, and the dgrad multiplies in fp32 BEFORE the conversion, like the staged path does.
              T* C2t = (ACT == ALPRO_ACT_GELU_SAVE_GRAD || ACT == ALPRO_ACT_MUL_SAVED) ? (T*)g.C2 + ((int64_t)tile * 8 + wave) * 8192 + le * 8 : nullptr;
              const int rows_here = g.M - mb;   // (> 0: waves without a valid row do not get here; >= 128 on all but the last tile row)
              auto drain = [&](T* base, int64_t ld, int u) {   // the staged 32 x 64 block -> four row pieces per lane
                u32x4 q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = *(const __attribute__((address_space(3))) u32x4*)(st8 + roff + ((j ^ (kgx & 3)) << 4));
                wave_lds_order();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  u32x4 o;
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const uint32_t a = (r & 2) ? q[j].y : q[j].x, b = (r & 2) ? q[j].w : q[j].z;   // column 2j / 2j + 1, rows (r & 2), (r & 2) + 1
                    o[j] = __builtin_amdgcn_perm(b, a, (r & 1) ? 0x07060302u : 0x05040100u);
                  }
                  if (32 * u + 4 * kgx + r < rows_here) __builtin_nontemporal_store(o, (u32x4*)(base + (int64_t)(32 * u + r) * ld));
                }
              };
              constexpr bool MULS = ACT == ALPRO_ACT_MUL_SAVED;
              // saved-factor passes in flight: two buffers of 16 registers; pass u + 2 is requested into pass u's buffer as soon as pass u's
              // products are formed, i.e. 1.5 passes (~1.5 us) ahead of its use (a third buffer spills accumulators at the path's entry)
              constexpr int PDU = 2, RINGU = MULS ? 2 : 1;
              u32x4 sring[RINGU][2][2];
              auto load_saved = [&](int u, u32x4(&ss)[2][2]) {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                  for (int j = 0; j < 2; ++j) ss[f][j] = __builtin_nontemporal_load((const u32x4*)(C2t + (2 * u + f) * 1024 + j * 512));
              };
              if constexpr (MULS) {
#pragma unroll
                for (int u = 0; u < PDU; ++u) load_saved(u, sring[u]);
              }
              const bool tiled = g.c2_tiled != 0;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                u32x2_t dpk[2][4];
                wave_lds_order();
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                  for (int nf = 0; nf < 4; ++nf) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = g.alpha * acc[2 * u + f][nf][r] + bcol[nf];
                    if constexpr (ACT == ALPRO_ACT_GELU_SAVE_GRAD) {
                      f32x2v y0, d0, y1, d1;
                      gelu_and_grad2((f32x2v){v[0], v[1]}, y0, d0);
                      gelu_and_grad2((f32x2v){v[2], v[3]}, y1, d1);
                      v[0] = y0.x; v[1] = y0.y; v[2] = y1.x; v[3] = y1.y;
                      dpk[f][nf] = (u32x2_t){pack2(d0.x, d0.y, (T*)0), pack2(d1.x, d1.y, (T*)0)};
                    } else if constexpr (ACT == ALPRO_ACT_GELU) {
                      const f32x2v y0 = gelu_fast2((f32x2v){v[0], v[1]}), y1 = gelu_fast2((f32x2v){v[2], v[3]});
                      v[0] = y0.x; v[1] = y0.y; v[2] = y1.x; v[3] = y1.y;
                    } else if constexpr (MULS) {
                      const u32x4& sv = sring[u % RINGU][f][nf >> 1];
                      float pre[4];
                      unpack_pair<T>((nf & 1) ? sv.z : sv.x, pre[0], pre[1]);
                      unpack_pair<T>((nf & 1) ? sv.w : sv.y, pre[2], pre[3]);
#pragma unroll
                      for (int r = 0; r < 4; ++r) v[r] *= pre[r];
                    } else {
#pragma unroll
                      for (int r = 0; r < 4; ++r) v[r] = apply_act<T, ACT>(v[r]);
                    }
                    *(__attribute__((address_space(3))) u32x2_t*)(st8 + woff + f * 2048 + nf * 128) = (u32x2_t){pack2(v[0], v[1], (T*)0), pack2(v[2], v[3], (T*)0)};
                  }
                wave_lds_order();
                if constexpr (MULS) {
                  if (u + PDU < 4) load_saved(u + PDU, sring[u % RINGU]);
                }
                drain(Cb, ldc, u);
                if constexpr (ACT == ALPRO_ACT_GELU_SAVE_GRAD) {
                  if (tiled) {
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                      for (int j = 0; j < 2; ++j)
                        __builtin_nontemporal_store(mk4(dpk[f][2 * j].x, dpk[f][2 * j].y, dpk[f][2 * j + 1].x, dpk[f][2 * j + 1].y), (u32x4*)(C2t + (2 * u + f) * 1024 + j * 512));
                  } else {
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                      for (int nf = 0; nf < 4; ++nf) *(__attribute__((address_space(3))) u32x2_t*)(st8 + woff + f * 2048 + nf * 128) = dpk[f][nf];
                    wave_lds_order();
                    drain(C2b, ldc2, u);
                  }
                }
                if (u == 0) early_ticket();
              }
            }
          } else {
          float bias8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) bias8[e] = g.bias ? g.bias[nb + (le & 7) * 8 + e] : 0.f;
          // The saved-factor rows (MUL_SAVED: gelu' of the forward, 16 bits) are fetched PD fragment rows ahead of their use.  A fragment row of
          // the epilogue takes ~0.5 us and an HBM round trip 1-2 us: with one row of run-ahead (round 4) every row waited for its loads -- the
          // whole gap between this dgrad (0.34 of peak in the step) and the plain 16-bit-output GEMM (0.40).  The ring lives in the registers the
          // K loop's fragments occupied (64 of them are free here): PD = 2 -> 3 x 8 registers (PD = 3 spills).
          constexpr int PD = 2, RING = PD + 1;
          u32x4 pring[RING][2];
          auto load_pre = [&](int mf, u32x4(&pp)[2]) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
              pp[p] = __builtin_nontemporal_load((const u32x4*)((const T*)g.C2 + (int64_t)(mb + mf * 16 + p * 8 + (le >> 3)) * g.ldc2 + nb + (le & 7) * 8));
          };
          if (READS_C2) {
#pragma unroll
            for (int mf = 0; mf < PD; ++mf)
              if (rows_ok(mf)) load_pre(mf, pring[mf]);
          }
          auto rows_loop = [&](auto res_tag) {
            constexpr bool RES = decltype(res_tag)::value;
#pragma unroll
            for (int mf = 0; mf < 8; ++mf) {
              if (!rows_ok(mf)) break;   // ragged last tile row: fragment rows at or beyond M are not stored (M % 16 == 0: launcher)
              if (READS_C2 && mf + PD < 8 && rows_ok(mf + PD)) load_pre(mf + PD, pring[(mf + PD) % RING]);
              stage_rows(mf);
              epi_rows16_c16<T, ACT, 2, 0, RES>(g, stage, mb + mf * 16, nb, le, bias8, READS_C2 ? pring[mf % RING] : nullptr);
              if constexpr (!RES && !READS_C2) {
                if (mf == 0) early_ticket();
              }
            }
          };
          if (g.residual) rows_loop(std::true_type{});   // (see epi_rows16_c16: no conditional vector load inside the passes)
          else rows_loop(std::false_type{});
          }
        }
      }
      // fp32 output (launcher: ACT none, identity map, no C2 / dropout): C = residual + row_scale * (alpha * acc + bias) -- the MLP's fc2 with its
      // fp32 residual (vit.py:212).  A staged fragment row is 16 rows x 16 float4; lane l finishes pieces l, l+64, l+128, l+192 = rows
      // (l >> 4) + 4j, columns 4 (l & 15) .. +3: whole 256-byte row segments per 16 lanes, the residual pieces of the NEXT fragment row in flight.
      if constexpr (MAP == ALPRO_MAP_IDENTITY && ACT == ALPRO_ACT_NONE) {
        if (g.c_dtype == ALPRO_F32) {
          const int c4 = (le & 15) * 4, r0e = le >> 4;
          float bias4[4];
          load_bias4(g, nb + c4, bias4);
          float* Cf = (float*)g.C;
          // Round 5: the residual rows run PD fragment rows ahead in a register ring and the row scales come WITHOUT a branch inside the
          // row loop.  (Round 4 held one row of run-ahead and a conditional row-scale load per row: the join behind it is closed with
          // s_waitcnt vmcnt(0), so every fragment row waited for the residual rows just requested AND for its predecessor's stores.)
          // The row scale of a fragment row: its 16 rows span at most two groups when row_scale_group >= 16 (the drop-path scale of fc2:
          // one value per clip of 1569 rows) -- one scalar division per fragment row, one or two scalar loads, a compare per row.
          constexpr int PD = 1, RING = PD + 1;   // (two rows ahead do not fit: 48 registers next to the 128 accumulators spill into the row loop)
          f32x4 ring[RING][4];
          auto load_res = [&](int mf, f32x4(&rr)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[j] = __builtin_nontemporal_load((const f32x4*)(g.residual + (int64_t)(mb + mf * 16 + r0e + 4 * j) * g.ldr + nb + c4));
          };
          auto f32_rows = [&](auto res_tag) {
            constexpr bool HAS_RES = decltype(res_tag)::value;
            if constexpr (HAS_RES) {
#pragma unroll
              for (int mf = 0; mf < PD; ++mf)
                if (rows_ok(mf)) load_res(mf, ring[mf]);
            }
#pragma unroll
            for (int mf = 0; mf < 8; ++mf) {
              if (!rows_ok(mf)) break;
              if constexpr (HAS_RES) {
                if (mf + PD < 8 && rows_ok(mf + PD)) load_res(mf + PD, ring[(mf + PD) % RING]);
              }
              float rs_lo = 1.0f, rs_hi = 1.0f;
              uint32_t edge = 0;   // first row (absolute, with m_off; rows < 2^31) of the second group
              if (g.row_scale) {   // scalar loads (wave-uniform addresses): no vector-memory join
                const uint32_t m0 = (uint32_t)g.m_off + (uint32_t)(mb + mf * 16);
                const uint32_t gi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(m0 / (uint32_t)g.row_scale_group));
                edge = (gi + 1) * (uint32_t)g.row_scale_group;
                rs_lo = sload_f32(g.row_scale, gi);
                rs_hi = sload_f32(g.row_scale, edge < (uint32_t)g.m_off + (uint32_t)g.M ? gi + 1 : gi);
              }
              stage_rows(mf);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int row = r0e + 4 * j;
                const int64_t m = mb + mf * 16 + row;
                const float4 a = *(const float4*)(stage + row * 64 + c4);
                const float rs = ((uint32_t)g.m_off + (uint32_t)m) >= edge ? rs_hi : rs_lo;   // (no row scale: both are 1)
                f32x4 v = {(g.alpha * a.x + bias4[0]) * rs, (g.alpha * a.y + bias4[1]) * rs, (g.alpha * a.z + bias4[2]) * rs, (g.alpha * a.w + bias4[3]) * rs};
                if constexpr (HAS_RES) v += ring[mf % RING][j];
                __builtin_nontemporal_store(v, (f32x4*)(Cf + m * g.ldc + nb + c4));
              }
              if constexpr (!HAS_RES) {
                if (mf == 0) early_ticket();
              }
            }
          };
          // (row_scale_group < 16 would need a scale per row: not a shape of this model -- the launcher keeps such descriptors off this kernel)
          if (g.residual) f32_rows(std::true_type{});
          else f32_rows(std::false_type{});
        }
      }
    }
    if (nxt_t < 0) break;
    cur = nxt;
    cur_t = nxt_t;
    nxt_t = __builtin_amdgcn_readfirstlane(mbox[0]);
    nxt = tile_base(nxt_t >= 0 ? nxt_t : cur_t);
    if (!early) {
      pending = sc.blk && nxt_t >= 0 && quota > 0;
      quota -= pending ? 1 : 0;
      ticket_issue_tk(wave == 0 && pending);
    }
  }
  // out of work: the run-ahead copies went into dead slots and must have landed before the stage buffers are filled again (or, at the end,
  // before the LDS belongs to someone else); steal() looks for other lists' tickets / unclaimed pairs next
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  cur_t = -1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int ACT, int MAP>
int launch_gemm_inst(const alpro_gemm_desc_t& g, hipStream_t st) {
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, ACT, MAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
    if constexpr (std::is_same<T, bf16_t>::value && ACT == ALPRO_ACT_NONE && MAP == ALPRO_MAP_IDENTITY) {
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
#ifdef ALPRO_ABLATIONS  // result-corrupting measurement variants: only in the tools/ build (python -m alpro_amd.build --ablations)
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);   // setprio variants
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<T, ACT, MAP, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES + EPI_BYTES);
#endif
    }
  });
  const int big_tiles = ((g.N + BN2 - 1) / BN2) * ((g.M + BM2 - 1) / BM2);
  const int force = get_option(OPT_GEMM_TILE);
  const int nk = (g.K * (int)sizeof(T)) / ROWB;
  // the persistent 256^2 kernel wins from ~160 tiles up (measured, tools/gemm_bert_bench.py: M=15168 N=768 = 180 tiles is 15-25 % faster than on the 128^2 kernel; M=2560 N=3072 = 120 tiles is not); its pipeline needs >= 2 K-tiles
  const bool use256 = nk >= 2 && (force ? force == 256 : big_tiles >= 160);
  if constexpr (sizeof(T) == 2 && MAP == ALPRO_MAP_IDENTITY) {
    // round 4: the 8-phase two-group schedule (gemm_nt256q_kernel) for the identity-map shapes; needs an even number >= 4 of 64-deep K-tiles
    // and 32-bit operand offsets.  gemm_kind 0 = the round-3 kernel (A/B), 1 = the 8-phase kernel
    static DeviceOnce attr_q;
    attr_q.run([&] {
      (void)hipFuncSetAttribute((const void*)gemm_nt256q_kernel<T, ACT, MAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE2_BYTES + EPI_BYTES);
    });
    // eligible: 16-bit output through the 16-byte-store epilogue, whole 256-column tiles, 128-byte-aligned operand rows, an even number >= 4 of
    // K-tiles; a ragged M (the ViT's B * 1569 token rows: M % 256 = 64) is split -- whole tiles here, the remaining rows on the 128 x 128 kernel
    // in a second launch -- when the epilogue does not index by absolute row (row scale, dropout)
    const int m_full = g.M / BM2 * BM2, m_rem = g.M - m_full;
    bool ragged_in_kernel = false;
    const bool take_q = q_kernel_takes(g, &ragged_in_kernel);
    ALPRO_CHECK(!g.c2_tiled || (take_q && c2_tiled_ok(g)), "alpro_gemm: c2_tiled is not available for this descriptor (M=%d N=%d K=%d act=%d): ask alpro_gemm_c2_tiled_rows first", g.M, g.N, g.K, g.act);
    if (take_q) {
      alpro_gemm_desc_t gq = g;
      if (!ragged_in_kernel) gq.M = m_full;
      // one workgroup per CU the launch may count on, whatever the tile count: the per-XCD lists are chunks of 32 tiles, and a workgroup
      // whose list (and, dynamic walk, every other list) is empty returns at once
      int grid = cu_budget(st);
      if (const int cap = get_option(OPT_GEMM_GRID)) grid = cap < grid ? (cap + 7) / 8 * 8 : grid;
      // Tail hand-over (round 6, option gemm_tail; VERDICT r3-r5 "the last partial round of the N = 768 projections").  With tiles = Q * grid + R
      // and a small R the last round keeps R workgroups busy for a whole tile time while the others idle -- at B = 32: the N = 768 projections
      // 588 / 591 tiles = 2.3 rounds, fc1 2352 = 9.19, fc2 588 (K = 3072).  A two-group 8-phase tile cannot be cut (a half tile takes as long as
      // a whole one: DESIGN.md), and split-K pays 256 KiB of fp32 partials per tile at K = 768.  What does work: the row panels that make up
      // the partial round go to the 128 x 128 kernel as a SECOND launch -- 4 R small tiles, two workgroups per CU, one short round behind the
      // Q full ones -- when R <= 0.4 grid (beyond that the small kernel's lower rate eats the saving) AND K >= 2048.  Measured (MI355X, fp16, B = 32,
      // tools/gemm_tail_ab.py, profiles/r6_gemm_tail_handover.txt): fc2 (K = 3072, 591 tiles) 274 -> 261 us (-4.7 %); the K = 768 projections
      // 74.4 -> 75.3 / 70.2 -> 71.2 us and fc1 (2364 tiles) 242 -> 239: the small kernel runs 304 tiles of 128 x 128 x 768 at ~300 TF/s, i.e. as
      // long as the round it replaces -- at K = 768 a tile is mostly epilogue and fill, and there the hand-over is NOT taken.  Same arithmetic per
      // element, different summation order inside the MFMAs: not bitwise equal to the unsplit launch (tests pin both against fp64),
      // bit-reproducible run to run.
      const int ntn_q = g.N / BN2, ntm_q = (gq.M + BM2 - 1) / BM2;
      const int tiles = ntn_q * ntm_q;
      int tail_panels = 0;
      if (get_option(OPT_GEMM_TAIL) == 1 && !g.c2_tiled && tiles > grid && g.K >= 2048) {
        const int R = tiles % grid;
        if (R > 0 && 5 * R <= 2 * grid) tail_panels = (R + ntn_q - 1) / ntn_q;
      }
      int64_t m_tail0 = -1;   // first row of what the second launch computes (-1: nothing)
      if (tail_panels > 0 && tail_panels < ntm_q) {
        m_tail0 = (int64_t)(ntm_q - tail_panels) * BM2;
        gq.M = (int)m_tail0;
      } else if (m_rem && !ragged_in_kernel) {
        m_tail0 = m_full;
      }
      TileSched sc;
      sc.blk = sc.prev = nullptr;
      sc.magic_ntn = magic_u32((uint32_t)(g.N / BN2));
      sc.epi = (uint32_t)get_option(OPT_GEMM_EPI);
      {
        SchedLaunch blocks(get_option(OPT_GEMM_SCHED) == 1 ? st : nullptr, get_option(OPT_GEMM_SCHED) == 1);   // (holds the stream's block pair until the launch is enqueued)
        sc.blk = blocks.cur;
        sc.prev = blocks.prev;
        hipLaunchKernelGGL((gemm_nt256q_kernel<T, ACT, MAP>), dim3(grid), dim3(NT2), 2 * STAGE2_BYTES + EPI_BYTES, st, gq, sc);
        blocks.commit(hipPeekAtLastError() == hipSuccess);
      }
      if (m_tail0 >= 0) {
        alpro_gemm_desc_t gr = g;
        gr.M = (int)(g.M - m_tail0);
        gr.A = (const char*)g.A + m_tail0 * g.lda * 2;
        gr.C = (char*)g.C + m_tail0 * g.ldc * (g.c_dtype == ALPRO_F32 ? 4 : 2);
        gr.m_off = g.m_off + m_tail0;
        if (g.C2) gr.C2 = (char*)g.C2 + m_tail0 * g.ldc2 * 2;
        if (g.residual) gr.residual = g.residual + m_tail0 * g.ldr;
        const int ntn = (gr.N + BN - 1) / BN, ntm = (gr.M + BM - 1) / BM;
        hipLaunchKernelGGL((gemm_nt_kernel<T, ACT, MAP>), dim3(ntn * ntm), dim3(NT), 4 * TILE_BYTES, st, gr);
      }
      return check_launch("alpro_gemm");
    }
  }
  if (use256) {
    const int cus = cu_budget(st);
    int grid = big_tiles < cus ? (big_tiles + 7) / 8 * 8 : cus;  // multiple of 8: the XCD-contiguous slot map must be a bijection
    if (const int cap = get_option(OPT_GEMM_GRID)) grid = cap < grid ? (cap + 7) / 8 * 8 : grid;  // tuning aid: cap the persistent grid
    const int tune = get_option(OPT_GEMM_TUNE);
    const int tail = get_option(OPT_GEMM_TAIL);
    if constexpr (std::is_same<T, bf16_t>::value && ACT == ALPRO_ACT_NONE && MAP == ALPRO_MAP_IDENTITY) {
      if (tune == 0) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 0>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
      if (tune == 2) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 2>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
#ifdef ALPRO_ABLATIONS
      if (tune == 3) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 3>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
      if (tune == 4) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 4>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
      if (tune == 12) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 12>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
      if (tune == 11) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 11>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
      if (tune == 10) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 10>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
      if (tune == 5) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 5>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
      if (tune == 6) { hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP, 6>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail); return check_launch("alpro_gemm"); }
#endif
    }
    hipLaunchKernelGGL((gemm_nt256p_kernel<T, ACT, MAP>), dim3(grid), dim3(NT2), 4 * TILE2_BYTES + EPI_BYTES, st, g, tail);
  } else {
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    hipLaunchKernelGGL((gemm_nt_kernel<T, ACT, MAP>), dim3(ntn * ntm), dim3(NT), 4 * TILE_BYTES, st, g);
  }
  return check_launch("alpro_gemm");
}

// activation and row map are compile-time (keeps each kernel's epilogue small); activations only combine
// with the identity map on this path (fc1 / BERT intermediate / MLM transform / mpm_head)
template <typename T>
int launch_gemm(const alpro_gemm_desc_t& g, hipStream_t st) {
#ifdef ALPRO_ISA_QUICK   // tools/isa_quick.sh: two instantiations per dtype instead of nine (register-pressure iterations on the 8-phase kernel; never the product build)
  return g.act == ALPRO_ACT_MUL_SAVED ? launch_gemm_inst<T, ALPRO_ACT_MUL_SAVED, ALPRO_MAP_IDENTITY>(g, st) : launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_IDENTITY>(g, st);
#else
  if (g.act != ALPRO_ACT_NONE) {
    if (g.map_mode != ALPRO_MAP_IDENTITY) {
      set_error("alpro_gemm: an activation cannot be combined with a row map");
      return ALPRO_ERR_INVALID;
    }
    if (g.act == ALPRO_ACT_GELU_BWD) return launch_gemm_inst<T, ALPRO_ACT_GELU_BWD, ALPRO_MAP_IDENTITY>(g, st);
    if (g.act == ALPRO_ACT_GELU_SAVE_GRAD) return launch_gemm_inst<T, ALPRO_ACT_GELU_SAVE_GRAD, ALPRO_MAP_IDENTITY>(g, st);
    if (g.act == ALPRO_ACT_MUL_SAVED) return launch_gemm_inst<T, ALPRO_ACT_MUL_SAVED, ALPRO_MAP_IDENTITY>(g, st);
    return g.act == ALPRO_ACT_GELU ? launch_gemm_inst<T, ALPRO_ACT_GELU, ALPRO_MAP_IDENTITY>(g, st)
                                   : launch_gemm_inst<T, ALPRO_ACT_RELU, ALPRO_MAP_IDENTITY>(g, st);
  }
  switch (g.map_mode) {
    case ALPRO_MAP_IDENTITY: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_IDENTITY>(g, st);
    case ALPRO_MAP_SKIP_CLS: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_SKIP_CLS>(g, st);
    case ALPRO_MAP_FRAME_TOKENS: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_FRAME_TOKENS>(g, st);
    default: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_PATCH_EMBED>(g, st);
  }
#endif
}
}  // namespace
}  // namespace alpro

namespace alpro {
namespace {
template <typename T>
int launch_gemm_batch(const alpro_gemm_desc_t* descs_dev, int njobs, int max_tiles, int with_residual_map, hipStream_t st) {
  (void)with_residual_map;
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_batch_kernel<T, ALPRO_ACT_NONE, ALPRO_MAP_IDENTITY>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
  });
  hipLaunchKernelGGL((gemm_nt_batch_kernel<T, ALPRO_ACT_NONE, ALPRO_MAP_IDENTITY>), dim3(max_tiles, njobs), dim3(NT), 4 * TILE_BYTES, st, descs_dev);
  return check_launch("alpro_gemm_batch");
}
}  // namespace
}  // namespace alpro

static int check_gemm_desc(const alpro_gemm_desc_t* d);

extern "C" int alpro_gemm_batch(const alpro_gemm_desc_t* descs_host, const alpro_gemm_desc_t* descs_device, int njobs, void* stream) {
  using namespace alpro;
  ALPRO_CHECK(descs_host && descs_device && njobs > 0, "alpro_gemm_batch: bad args");
  int max_tiles = 0;
  for (int i = 0; i < njobs; ++i) {
    const alpro_gemm_desc_t* d = descs_host + i;
    if (const int rc = check_gemm_desc(d)) return rc;
    ALPRO_CHECK(d->dtype == descs_host[0].dtype, "alpro_gemm_batch: all jobs must share the operand dtype");
    ALPRO_CHECK(d->act == ALPRO_ACT_NONE && d->map_mode == ALPRO_MAP_IDENTITY && !d->drop_seed && !d->bias2 && !d->C2,
                "alpro_gemm_batch: plain epilogues only (bias, alpha, row scale, fp32 residual)");
    const int t = ((d->N + BN - 1) / BN) * ((d->M + BM - 1) / BM);
    max_tiles = t > max_tiles ? t : max_tiles;
  }
  ALPRO_DISPATCH_DTYPE(descs_host[0].dtype, T, return launch_gemm_batch<T>(descs_device, njobs, max_tiles, 0, (hipStream_t)stream));
  return ALPRO_OK;
}

static int check_gemm_desc(const alpro_gemm_desc_t* d) {
  using namespace alpro;
  ALPRO_CHECK(d && d->A && d->W && d->C, "alpro_gemm: null operand");
  ALPRO_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "alpro_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  const int esz = d->dtype == ALPRO_F32 ? 4 : 2;
  ALPRO_CHECK((d->K * esz) % 128 == 0, "alpro_gemm: K=%d must be a multiple of %d for dtype %d", d->K, 128 / esz, d->dtype);
  ALPRO_CHECK((d->lda * esz) % 16 == 0 && (d->ldw * esz) % 16 == 0, "alpro_gemm: lda/ldw must keep rows 16-byte aligned");
  ALPRO_CHECK(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->W % 16) == 0, "alpro_gemm: A/W must be 16-byte aligned");
  ALPRO_CHECK(d->c_dtype == d->dtype || d->c_dtype == ALPRO_F32, "alpro_gemm: c_dtype must be dtype or F32");
  ALPRO_CHECK(d->map_mode >= 0 && d->map_mode <= 3, "alpro_gemm: bad map_mode %d", d->map_mode);
  ALPRO_CHECK(d->act >= 0 && d->act <= ALPRO_ACT_MUL_SAVED, "alpro_gemm: bad act %d", d->act);
  ALPRO_CHECK(!d->bias2 || d->map_mode == ALPRO_MAP_SKIP_CLS, "alpro_gemm: bias2 is only defined under the SKIP_CLS map");
  ALPRO_CHECK((d->act != ALPRO_ACT_GELU_BWD && d->act != ALPRO_ACT_MUL_SAVED && d->act != ALPRO_ACT_GELU_SAVE_GRAD) || (d->C2 && d->N % 8 == 0 && (d->c2_tiled || d->ldc2 % 8 == 0)),
              "alpro_gemm: GELU_BWD / MUL_SAVED / GELU_SAVE_GRAD need the C2 buffer (N, ldc2 multiples of 8)");
  ALPRO_CHECK(!d->drop_seed || (d->map_mode == ALPRO_MAP_IDENTITY && d->drop_p > 0.f && d->drop_p < 1.f), "alpro_gemm: dropout needs the identity map and 0 < p < 1");
  ALPRO_CHECK(!d->C2 || (d->N % 4 == 0 && (d->c2_tiled || d->ldc2 % 4 == 0) && d->ldc % 4 == 0), "alpro_gemm: C2 needs N, ldc, ldc2 multiples of 4");
  ALPRO_CHECK(!d->c2_tiled || d->act == ALPRO_ACT_GELU_SAVE_GRAD || d->act == ALPRO_ACT_MUL_SAVED, "alpro_gemm: c2_tiled is for GELU_SAVE_GRAD / MUL_SAVED only");
  ALPRO_CHECK(d->map_mode != ALPRO_MAP_FRAME_TOKENS || d->side, "alpro_gemm: FRAME_TOKENS needs a side buffer");
  ALPRO_CHECK(!d->row_scale || d->row_scale_group > 0, "alpro_gemm: row_scale_group must be > 0");
  return ALPRO_OK;
}

extern "C" int64_t alpro_gemm_c2_tiled_rows(int64_t M, int64_t N, int64_t K, int dtype) {
  using namespace alpro;
  if (M <= 0 || N <= 0 || K <= 0 || M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff || (dtype != ALPRO_F16 && dtype != ALPRO_BF16)) return 0;
  alpro_gemm_desc_t g = {};
  g.A = g.W = (const void*)(uintptr_t)256;   // (never dereferenced: alignment only)
  g.C = g.C2 = (void*)(uintptr_t)256;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = g.ldw = K; g.ldc = N;
  g.dtype = g.c_dtype = dtype;
  g.act = ALPRO_ACT_GELU_SAVE_GRAD;
  g.c2_tiled = 1;
  bool ragged = false;
  return (q_kernel_takes(g, &ragged) && c2_tiled_ok(g)) ? (M + 255) / 256 * 256 : 0;
}

extern "C" int alpro_gemm(const alpro_gemm_desc_t* d, void* stream) {
  using namespace alpro;
  if (const int rc = check_gemm_desc(d)) return rc;
#ifdef ALPRO_ISA_QUICK
  return launch_gemm<f16_t>(*d, (hipStream_t)stream);
#else
  ALPRO_DISPATCH_DTYPE(d->dtype, T, return launch_gemm<T>(*d, (hipStream_t)stream));
  return ALPRO_OK;
#endif
}
