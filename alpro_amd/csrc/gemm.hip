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
__device__ __forceinline__ RowDst map_row(int mode, int p0, int p1, int m) {
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

template <typename T>
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

  // ---- epilogue: accumulators -> wave-private LDS (reusing the staging buffers) -> row-wise 16-byte I/O ----
  // (the trailing __syncthreads of the K loop guarantees nobody still reads the staging tiles)
  float* stage = (float*)(smem + wave * (64 * 64 * 4));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[(i * 32 + acc_row(r, lane)) * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int c4 = (lane & 15) * 4;
  const int n = n0 + wc * 64 + c4;
  const bool vec_ok = (n + 3 < g.N) && ((g.ldc & 3) == 0) && (!g.residual || (g.ldr & 3) == 0) && ((g.ld_side & 3) == 0);
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < g.N) bias[e] = g.bias[n + e];
  }
#pragma unroll 4
  for (int pass = 0; pass < 16; ++pass) {
    const int row = pass * 4 + (lane >> 4);
    const int m = m0 + wr * 64 + row;
    if (m < g.M && n < g.N) {
      const float4 a = *(const float4*)(stage + row * 64 + c4);
      float v[4] = {a.x, a.y, a.z, a.w};
      const float rs = g.row_scale ? g.row_scale[m / g.row_scale_group] : 1.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = g.alpha * v[e] + bias[e];
        if (g.act == ALPRO_ACT_GELU) x = gelu_erf(x);
        else if (g.act == ALPRO_ACT_RELU) x = fmaxf(x, 0.f);
        v[e] = x * rs;
      }
      const RowDst d = map_row(g.map_mode, g.map_p0, g.map_p1, m);
      if (d.side) {
        float* dst = g.side + d.out * g.ld_side + n;
        if (vec_ok) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < g.N) dst[e] = v[e];
        }
      } else if (vec_ok) {
        if (g.residual) {
          const float4 rr = *(const float4*)(g.residual + d.res * g.ldr + n);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (g.c_dtype == ALPRO_F32) {
          *(float4*)((float*)g.C + d.out * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else if constexpr (sizeof(T) == 2) {
          u32x2 u;
          u.x = pack2(v[0], v[1], (T*)0);
          u.y = pack2(v[2], v[3], (T*)0);
          *(u32x2*)((T*)g.C + d.out * g.ldc + n) = u;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < g.N) {
            float x = v[e];
            if (g.residual) x += g.residual[d.res * g.ldr + n + e];
            store_c<T>(g.C, g.c_dtype, d.out * g.ldc + n + e, x);
          }
      }
    }
  }
}

template <typename T>
int launch_gemm(const alpro_gemm_desc_t& g, hipStream_t st) {
  const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
  const size_t lds = 4 * TILE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_nt_kernel<T>, dim3(ntn * ntm), dim3(NT), lds, st, g);
  return check_launch("alpro_gemm");
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
  ALPRO_CHECK(d->map_mode != ALPRO_MAP_FRAME_TOKENS || d->side, "alpro_gemm: FRAME_TOKENS needs a side buffer");
  ALPRO_CHECK(!d->row_scale || d->row_scale_group > 0, "alpro_gemm: row_scale_group must be > 0");
  ALPRO_DISPATCH_DTYPE(d->dtype, T, return launch_gemm<T>(*d, (hipStream_t)stream));
  return ALPRO_OK;
}
