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


// Epilogue of one 64x64 wave sub-tile: accumulators -> wave-private LDS -> row-wise 16-byte I/O, so that
// bias / activation / drop-path scale / residual / row-map work on 4 consecutive columns of ONE row per lane.
__device__ __forceinline__ void stage_acc(float* stage, const f32x16& a00, const f32x16& a01, const f32x16& a10, const f32x16& a11, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane), col = lane & 31;
    stage[row * 64 + col] = a00[r];
    stage[row * 64 + 32 + col] = a01[r];
    stage[(32 + row) * 64 + col] = a10[r];
    stage[(32 + row) * 64 + 32 + col] = a11[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int ACT> __device__ __forceinline__ float apply_act(float x) {
  if (ACT == ALPRO_ACT_GELU) return gelu_fast<T>(x);
  if (ACT == ALPRO_ACT_RELU) return fmaxf(x, 0.f);
  return x;
}

// Rows of the staged 64x64 tile.  Two sweeps (runtime loop: keeps the code small enough for the I-cache)
// of 8 row-passes; phase A issues every residual load of the sweep (the accumulator registers are dead by
// now), phase B does the math and the stores -- otherwise the epilogue is a chain of dependent HBM round trips.
template <typename T, int ACT, int MAP>
__device__ __forceinline__ void epilogue_rows(const alpro_gemm_desc_t& g, const float* stage, int m_base, int n_base, int lane) {
  const int c4 = (lane & 15) * 4;
  const int n = n_base + c4;
  const bool vec_ok = (n + 3 < g.N) && ((g.ldc & 3) == 0) && (!g.residual || (g.ldr & 3) == 0) && ((g.ld_side & 3) == 0);
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < g.N) bias[e] = g.bias[n + e];
  }
#pragma unroll 1
  for (int sweep = 0; sweep < 2; ++sweep) {
    float4 rr[8];
    int64_t orow[8];
    bool live[8], side[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int m = m_base + (sweep * 8 + p) * 4 + (lane >> 4);
      live[p] = m < g.M && n < g.N;
      const RowDst d = map_row<MAP>(g.map_p0, g.map_p1, live[p] ? m : 0);
      orow[p] = d.out;
      side[p] = d.side;
      rr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live[p] && g.residual && !d.side) {
        const float* rp = g.residual + d.res * g.ldr + n;
        if (vec_ok) rr[p] = *(const float4*)rp;
        else {
          rr[p].x = rp[0];
          if (n + 1 < g.N) rr[p].y = rp[1];
          if (n + 2 < g.N) rr[p].z = rp[2];
          if (n + 3 < g.N) rr[p].w = rp[3];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = (sweep * 8 + p) * 4 + (lane >> 4);
      const int m = m_base + row;
      if (live[p]) {
        const float4 a = *(const float4*)(stage + row * 64 + c4);
        float v[4] = {a.x, a.y, a.z, a.w};
        const float res[4] = {rr[p].x, rr[p].y, rr[p].z, rr[p].w};
        const float rs = g.row_scale ? g.row_scale[m / g.row_scale_group] : 1.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = g.alpha * v[e] + bias[e];
        if (ACT != ALPRO_ACT_NONE && g.C2 && vec_ok) {  // pre-activation copy for the activation's backward
          if constexpr (sizeof(T) == 2) {
            u32x2 u;
            u.x = pack2(v[0], v[1], (T*)0);
            u.y = pack2(v[2], v[3], (T*)0);
            *(u32x2*)((T*)g.C2 + orow[p] * g.ldc2 + n) = u;
          } else {
            *(float4*)((float*)g.C2 + orow[p] * g.ldc2 + n) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<T, ACT>(v[e]) * rs + res[e];
        if (side[p]) {
          float* dst = g.side + orow[p] * g.ld_side + n;
          if (vec_ok) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < g.N) dst[e] = v[e];
          }
        } else if (vec_ok) {
          if (g.c_dtype == ALPRO_F32) {
            *(float4*)((float*)g.C + orow[p] * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else if constexpr (sizeof(T) == 2) {
            u32x2 u;
            u.x = pack2(v[0], v[1], (T*)0);
            u.y = pack2(v[2], v[3], (T*)0);
            *(u32x2*)((T*)g.C + orow[p] * g.ldc + n) = u;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < g.N) store_c<T>(g.C, g.c_dtype, orow[p] * g.ldc + n + e, v[e]);
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <typename T, int ACT, int MAP>
__global__ __launch_bounds__(NT, 2) void gemm_nt_kernel(const alpro_gemm_desc_t g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
  const int nblk = ntn * ntm;
  // XCD-aware, bijective remap of blockIdx -> logical tile
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
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
  stage_acc(stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], lane);
  epilogue_rows<T, ACT, MAP>(g, stage, m0 + wr * 64, n0 + wc * 64, lane);
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4, each 128x64 = 4x2 MFMA accumulators), K-tile 128 bytes, two LDS stages of
// 64 KiB filled by global_load_lds_dwordx4 (no VGPR round trip, no ds_write).  The LDS image written by
// the DMA is lane-linear (1 KiB = 8 rows per wave instruction), so the bank swizzle is applied to the
// per-lane SOURCE chunk and undone by the same XOR on the fragment read (linear dest + swizzled source).
constexpr int BM2 = 256, BN2 = 256, NT2 = 512;
constexpr int TILE2_BYTES = BM2 * ROWB;  // 32 KiB per operand per stage

template <typename T, int ACT, int MAP>
__global__ __launch_bounds__(NT2, 2) void gemm_nt256_kernel(const alpro_gemm_desc_t g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (g.N + BN2 - 1) / BN2, ntm = (g.M + BM2 - 1) / BM2;
  const int nblk = ntn * ntm;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int64_t lda_b = g.lda * (int64_t)sizeof(T), ldw_b = g.ldw * (int64_t)sizeof(T);

  // DMA pieces: piece p = 8 rows (1 KiB); wave w moves pieces w, w+8, w+16, w+24 of A and of W per K-tile
  const char* a_src[4];
  const char* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 8 * i) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);  // source chunk that belongs in LDS slot (row, lane & 7)
    a_src[i] = (const char*)g.A + min(m0 + row, g.M - 1) * lda_b + ch * 16;
    w_src[i] = (const char*)g.W + min(n0 + row, g.N - 1) * ldw_b + ch * 16;
  }
  int a_row[4], b_row[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_row[i] = wr * 128 + i * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 2; ++j) b_row[j] = wc * 64 + j * 32 + (lane & 31);
  const int khalf = lane >> 5;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (g.K * (int)sizeof(T)) / ROWB;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  auto stage_tile = [&](int kt, int buf) {
    char* dA = smem + buf * 2 * TILE2_BYTES;
    char* dW = dA + TILE2_BYTES;
    const int64_t ko = (int64_t)kt * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_ptr)(a_src[i] + ko), (lds_ptr)(dA + (wave + 8 * i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr)(w_src[i] + ko), (lds_ptr)(dW + (wave + 8 * i) * 1024), 16, 0, 0);
    }
  };
  stage_tile(0, 0);
  __syncthreads();  // (drains vmcnt: the DMA of tile 0 has landed)

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage_tile(kt + 1, cur ^ 1);
    const char* cA = smem + cur * 2 * TILE2_BYTES;
    const char* cW = cA + TILE2_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[4], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *(const u32x4*)(cW + lds_off(b_row[j], 2 * s + khalf));
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const u32x4*)(cA + lds_off(a_row[i], 2 * s + khalf));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_chunk<T>(acc[i][j], fa[i], fb[j]);
    }
    __syncthreads();  // next tile landed (vmcnt(0)) and everyone is done reading this one
    cur ^= 1;
  }

  float* stage = (float*)(smem + wave * (64 * 64 * 4));
  stage_acc(stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], lane);
  epilogue_rows<T, ACT, MAP>(g, stage, m0 + wr * 128, n0 + wc * 64, lane);
  stage_acc(stage, acc[2][0], acc[2][1], acc[3][0], acc[3][1], lane);
  epilogue_rows<T, ACT, MAP>(g, stage, m0 + wr * 128 + 64, n0 + wc * 64, lane);
}

template <typename T, int ACT, int MAP>
int launch_gemm_inst(const alpro_gemm_desc_t& g, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, ACT, MAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, ACT, MAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES);
    attr_set = true;
  }
  const int big_tiles = ((g.N + BN2 - 1) / BN2) * ((g.M + BM2 - 1) / BM2);
  const char* force = getenv("ALPRO_GEMM_TILE");
  const bool use256 = force ? atoi(force) == 256 : big_tiles >= 256;  // at least one full wave of 256^2 tiles on 256 CUs
  if (use256) {
    hipLaunchKernelGGL((gemm_nt256_kernel<T, ACT, MAP>), dim3(big_tiles), dim3(NT2), 4 * TILE2_BYTES, st, g);
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
  if (g.act != ALPRO_ACT_NONE) {
    if (g.map_mode != ALPRO_MAP_IDENTITY) {
      set_error("alpro_gemm: an activation cannot be combined with a row map");
      return ALPRO_ERR_INVALID;
    }
    return g.act == ALPRO_ACT_GELU ? launch_gemm_inst<T, ALPRO_ACT_GELU, ALPRO_MAP_IDENTITY>(g, st)
                                   : launch_gemm_inst<T, ALPRO_ACT_RELU, ALPRO_MAP_IDENTITY>(g, st);
  }
  switch (g.map_mode) {
    case ALPRO_MAP_IDENTITY: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_IDENTITY>(g, st);
    case ALPRO_MAP_SKIP_CLS: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_SKIP_CLS>(g, st);
    case ALPRO_MAP_FRAME_TOKENS: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_FRAME_TOKENS>(g, st);
    default: return launch_gemm_inst<T, ALPRO_ACT_NONE, ALPRO_MAP_PATCH_EMBED>(g, st);
  }
}
}  // namespace
}  // namespace alpro

extern "C" int alpro_gemm(const alpro_gemm_desc_t* d, void* stream) {
  using namespace alpro;
  ALPRO_CHECK(d && d->A && d->W && d->C, "alpro_gemm: null operand");
  ALPRO_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "alpro_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  const int esz = d->dtype == ALPRO_F32 ? 4 : 2;
  ALPRO_CHECK((d->K * esz) % 128 == 0, "alpro_gemm: K=%d must be a multiple of %d for dtype %d", d->K, 128 / esz, d->dtype);
  ALPRO_CHECK((d->lda * esz) % 16 == 0 && (d->ldw * esz) % 16 == 0, "alpro_gemm: lda/ldw must keep rows 16-byte aligned");
  ALPRO_CHECK(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->W % 16) == 0, "alpro_gemm: A/W must be 16-byte aligned");
  ALPRO_CHECK(d->c_dtype == d->dtype || d->c_dtype == ALPRO_F32, "alpro_gemm: c_dtype must be dtype or F32");
  ALPRO_CHECK(d->map_mode >= 0 && d->map_mode <= 3, "alpro_gemm: bad map_mode %d", d->map_mode);
  ALPRO_CHECK(d->act >= 0 && d->act <= 2, "alpro_gemm: bad act %d", d->act);
  ALPRO_CHECK(!d->C2 || (d->N % 4 == 0 && d->ldc2 % 4 == 0 && d->ldc % 4 == 0), "alpro_gemm: C2 needs N, ldc, ldc2 multiples of 4");
  ALPRO_CHECK(d->map_mode != ALPRO_MAP_FRAME_TOKENS || d->side, "alpro_gemm: FRAME_TOKENS needs a side buffer");
  ALPRO_CHECK(!d->row_scale || d->row_scale_group > 0, "alpro_gemm: row_scale_group must be > 0");
  ALPRO_DISPATCH_DTYPE(d->dtype, T, return launch_gemm<T>(*d, (hipStream_t)stream));
  return ALPRO_OK;
}
