// Weight-gradient GEMM for gfx950:  C[N, K] += A[M, N]^T * B[M, K]   (A = dY, B = X, both row-major in a 16-bit
// dtype, contraction over the TOKEN dimension M, fp32 accumulation, fp32 atomic accumulate into C = param.grad).
//
// Both operands are read in their natural layout -- no transposed copies in HBM:
//  * 256 x 256 output tile, 8 waves (2 x 4, 128 x 64 each); the token range is consumed in 32-token stages through a
//    4-stage LDS ring (two (32 tokens x 256 columns) images per stage, 128 KiB) filled by global_load_lds_dwordx4 issued
//    from inline asm, three stages in flight, tracked with exact s_waitcnt vmcnt counts (every wave issues 4 copies per
//    step, zero-page dummies past the end); copies are placed between the MFMAs and the two waves of a SIMD use
//    alternating slots; fragments are double buffered across a mid-step barrier;
//  * MFMA operands need 8 consecutive tokens of ONE column per lane: ds_read_b64_tr_b16 gathers 4 tokens x 16
//    columns per 16-lane group (semantics: tools/probe_tr.hip); the 16-byte chunk index is XOR-ed with
//    (token & 3) << 2 on the DMA source side so the four token rows of a gather sit in different 64-byte windows;
//  * split over M: wgrad outputs are tiny (768..3072 x 768) while M is ~10^5, so the token range is cut into slices;
//    slice s runs on XCD s % 8 (1-D grid decoded by hand, tools/probe_xcd.hip) so each token row is fetched from HBM
//    once and re-used by the slice's other tiles out of that XCD's L2; the slice count is chosen against the measured
//    fixed cost of a workgroup (~26 us, mostly the 256 KiB atomic epilogue); partial tiles are combined with hardware
//    fp32 atomics, which is also what "accumulate into .grad" needs;
//  * optional bias gradient: the waves of the first k-tile column sum their dY fragments on the VALU (colsum).
// Rows >= M / columns >= N,K read a zero page, so no operand needs padding.
#include "common.hpp"

namespace alpro {
namespace {

constexpr int TM = 32;          // tokens per stage
constexpr int NSTAGE = 4;       // stages in LDS (3 in flight while one is consumed)
constexpr int TW = 256;         // tile width (columns of A -> rows of C; columns of B -> columns of C)
constexpr int ROW_BYTES = TW * 2;
constexpr int IMG_BYTES = TM * ROW_BYTES;  // 16 KiB
constexpr int NT3 = 512;

__device__ u32x4 g_zero_page[4];

typedef short s16x4 __attribute__((ext_vector_type(4)));

// 4 consecutive tokens (m0 multiple of 4) of column `col` (tile-local) for this lane: lane p of each 16-lane group
// addresses token m0 + (p >> 2), 8-byte piece (p & 3) of the group's 16-column block.
__device__ __forceinline__ u32x2 tr4(const char* img, int m0, int colblk16, int lane) {
  const int p = lane & 15;
  const int row = m0 + (p >> 2);
  const int chunk = (colblk16 << 1) | ((p >> 1) & 1);
  const char* a = img + row * ROW_BYTES + ((chunk ^ ((row & 3) << 2)) << 4) + ((p & 1) << 3);
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
  return __builtin_bit_cast(u32x2, r);
}
// 8-token operand chunk of MFMA k-step ks for the 32 columns [col0, col0+32): tokens ks*16 + 4g + {0..3} and + 8
__device__ __forceinline__ u32x4 frag8(const char* img, int ks, int col0, int lane) {
  const int g = lane >> 5;
  const int blk = (col0 >> 4) + ((lane >> 4) & 1);
  const u32x2 a = tr4(img, ks * 16 + 4 * g, blk, lane);
  const u32x2 b = tr4(img, ks * 16 + 8 + 4 * g, blk, lane);
  const uint32_t ax = a.x, ay = a.y, bx = b.x, by = b.y;
  return mk4(ax, ay, bx, by);
}

template <typename T>
__global__ __launch_bounds__(NT3, 1) void gemm_tn_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K, int steps_per_split,
                                                         int tn_cnt, int tk_cnt, float* __restrict__ colsum) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // XCD-aware decode of the 1-D grid: all (n-tile, k-tile) workgroups of one token slice run on ONE XCD (blockIdx % 8)
  // back to back, so each token row of dY and X is fetched from HBM once and re-used out of that XCD's L2 by the
  // other tiles of the slice (without this the operands are re-read N/256 resp. K/256 times).
  const int tiles = tn_cnt * tk_cnt;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int slice = (idx / tiles) * 8 + xcd;
  const int tile = idx - (idx / tiles) * tiles;
  const int n0 = (tile % tn_cnt) * TW, k0 = (tile / tn_cnt) * TW;
  const int total_steps = (M + TM - 1) / TM;
  const int s0 = slice * steps_per_split;
  const int s1 = min(s0 + steps_per_split, total_steps);
  if (s0 >= s1) return;

  // DMA pieces: 1 KiB = 2 token rows x 512 B; wave w moves pieces w and w+8 of each image per stage.
  // LDS slot (row, chunk') receives source chunk chunk' ^ ((row & 3) << 2).  Each lane keeps a byte cursor per piece
  // that advances by one stage (TM rows) per issue; lanes whose column is outside the matrix, and rows past the end
  // of the slice, read the zero page instead (cursor stride 0), so the loop body has no 64-bit address arithmetic
  // beyond one add per cursor.
  const char* zero = (const char*)g_zero_page;
  const int rows_end = min(s1 * TM, M);
  const char* curA[2];
  const char* curB[2];
  int mrow[2];
  int64_t strA[2], strB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + 8 * i) * 2 + (lane >> 5);
    const int col = ((lane & 31) ^ ((row & 3) << 2)) * 8;
    const int64_t m = (int64_t)s0 * TM + row;
    const bool ca = n0 + col < N, cb = k0 + col < K;
    mrow[i] = s0 * TM + row;
    curA[i] = ca ? (const char*)(A + m * lda + n0 + col) : zero;
    curB[i] = cb ? (const char*)(B + m * ldb + k0 + col) : zero;
    strA[i] = ca ? (int64_t)TM * lda * (int64_t)sizeof(T) : 0;
    strB[i] = cb ? (int64_t)TM * ldb * (int64_t)sizeof(T) : 0;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lds_piece[2] = {lds_base + wave * 1024u, lds_base + (wave + 8) * 1024u};
  // copy d = 0..3 of a stage: (A, piece 0), (B, piece 0), (A, piece 1), (B, piece 1) into stage buffer `buf`
  auto copy = [&](int d, int buf) {
    const int i = d >> 1;
    const char* src = (d & 1) ? curB[i] : curA[i];
    src = mrow[i] < rows_end ? src : zero;
    dma16(src, __builtin_amdgcn_readfirstlane(lds_piece[i] + buf * 2 * IMG_BYTES + (d & 1) * IMG_BYTES));
  };
  auto advance = [&](int i) {  // piece i's cursors -> next stage
    curA[i] += strA[i];
    curB[i] += strB[i];
    mrow[i] += TM;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline (4 stage buffers, fragments double-buffered in registers).  Step st:
  //   first half : F1 <- LDS (stage st, tokens 16..31);  8 MFMAs on F0, with copies 2,3 of stage st+3 in the gaps
  //   middle     : F1 landed (lgkmcnt 0) -> this wave is done reading stage st;  stage st+1 landed (vmcnt <= 8:
  //                stages st+2, st+3 stay in flight);  barrier
  //   second half: F0 <- LDS (stage st+1, tokens 0..15);  8 MFMAs on F1, with copies 0,1 of stage st+4 in the gaps
  //                (they overwrite stage st's buffer, which every wave has finished reading at the barrier)
  // so no MFMA waits on an LDS read issued in the same half, and every wave issues exactly 4 copies per step
  // (out-of-range stages read the zero page), which is what makes the vmcnt arithmetic exact.
  // A copy stalls its wave for ~60-150 issue cycles (the CU's address unit takes 1 KiB per ~16 clk): the two waves
  // that share a SIMD (w, w+4) use alternating slots so that the partner's MFMAs cover it.
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int d = 0; d < 4; ++d) copy(d, p);
    advance(0);
    advance(1);
  }
  copy(0, 3);
  copy(1, 3);  // piece 0 of stage s0+3; piece 1 follows in the first half of step s0
  advance(0);
  const int pos = wave >> 2;
  auto load_frags = [&](u32x4* fa, u32x4* fb, const char* cA, const char* cB, int ks) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = frag8(cB, ks, wc * 64 + j * 32, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag8(cA, ks, wr * 128 + i * 32, lane);
  };
  // Bias gradient for free: colsum[n] += sum over tokens of A[m, n].  The A fragments already sit in registers (lane = one
  // column, 8 tokens per fragment), so the waves of the first k-tile column (k0 == 0, wc == 0) add them up on the VALU
  // while the MFMA pipe runs -- the separate pass over dY that torch / alpro_colsum_acc would make never happens.
  const bool do_cs = colsum != nullptr && k0 == 0 && wc == 0;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  auto add_cols = [&](const u32x4* fa) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[8];
      unpack_chunk<T>(fa[i], f);
      cs[i] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    }
  };
  u32x4 fa0[4], fb0[2], fa1[4], fb1[2];
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_frags(fa0, fb0, smem, smem + IMG_BYTES, 0);
  for (int st = s0; st < s1; ++st) {
    const int cur = (st - s0) & (NSTAGE - 1);
    const int b3 = (cur + 3) & (NSTAGE - 1), b1 = (cur + 1) & (NSTAGE - 1);
    const char* cA = smem + cur * 2 * IMG_BYTES;
    load_frags(fa1, fb1, cA, cA + IMG_BYTES, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        mma_chunk<T>(acc[i][j], fa0[i], fb0[j]);
        if (j == 1 && (i & 1) == pos) copy(2 + (i >> 1), b3);  // piece 1 (A, B) of stage st+3
      }
    if (do_cs) add_cols(fa0);
    advance(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* nA = smem + b1 * 2 * IMG_BYTES;
    load_frags(fa0, fb0, nA, nA + IMG_BYTES, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        mma_chunk<T>(acc[i][j], fa1[i], fb1[j]);
        if (j == 1 && (i & 1) == pos) copy(i >> 1, cur);  // piece 0 (A, B) of stage st+4
      }
    if (do_cs) add_cols(fa1);
    advance(0);
  }
  if (do_cs) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = cs[i] + __shfl_xor(cs[i], 32, 64);  // the two token halves of the fragment
      const int n = n0 + wr * 128 + i * 32 + (lane & 31);
      if (lane < 32 && n < N) unsafeAtomicAdd(colsum + n, t);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail copies (zero page) before the wave exits
  // C[n, k] += acc: lane owns column k = k0 + wc*64 + j*32 + (lane & 31); hardware fp32 atomics
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + wc * 64 + j * 32 + (lane & 31);
    if (k < K) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wr * 128 + i * 32 + acc_row(r, lane);
          if (n < N) unsafeAtomicAdd(C + (int64_t)n * ldc + k, acc[i][j][r]);
        }
    }
  }
}

// colsum[n] += sum_m A[m, n]: bias gradients.  One pass over dY: a wave reads 64 consecutive 16-byte chunks of a row
// (1 KiB), the 4 waves of a workgroup take rows r, r+4, ... of a 256-row strip with 8 independent loads in flight,
// then a 4-way LDS reduction and one fp32 atomic per column per strip.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ A, int64_t lda, float* __restrict__ out, int M, int N) {
  constexpr int E = Chunk<T>::N;
  __shared__ float red[4][64 * E];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;           // 16-byte chunk of the row
  const bool cv = c * E < N;
  const int m0 = blockIdx.y * 256, m1 = min(m0 + 256, M);
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  if (cv) {
    const T* base = A + c * E;
    int m = m0 + w;
    for (; m + 28 < m1; m += 32) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const u32x4*)(base + (int64_t)(m + 4 * u) * lda);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float f[E];
        unpack_chunk<T>(v[u], f);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += f[e];
      }
    }
    for (; m < m1; m += 4) {
      float f[E];
      unpack_chunk<T>(*(const u32x4*)(base + (int64_t)m * lda), f);
#pragma unroll
      for (int e = 0; e < E; ++e) s[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) red[w][lane * E + e] = s[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * E; i += 256) {
    const int col = blockIdx.x * 64 * E + i;
    if (col < N) unsafeAtomicAdd(out + col, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_gemm_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int dtype, int M, int N,
                                 int K, float* colsum, void* stream) {
  ALPRO_CHECK(A && B && C && M > 0 && N > 0 && K > 0, "alpro_gemm_tn_acc: bad args");
  ALPRO_CHECK(dtype == ALPRO_BF16 || dtype == ALPRO_F16, "alpro_gemm_tn_acc: 16-bit operands only (fp32 mode uses alpro_transpose + alpro_gemm)");
  ALPRO_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= (N + 7) / 8 * 8 && ldb >= (K + 7) / 8 * 8,
              "alpro_gemm_tn_acc: lda/ldb must be multiples of 8 covering N/K rounded up to 8 (16-byte chunks are read whole)");
  ALPRO_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "alpro_gemm_tn_acc: operands must be 16-byte aligned");
  const int tn = (N + TW - 1) / TW, tk = (K + TW - 1) / TW;
  const int tiles = tn * tk;
  const int total_steps = (M + TM - 1) / TM;
  // Token slices: a multiple of 8 (slice s runs on XCD s % 8 -- see the kernel), s8 per XCD.  Pick the s8 whose
  // workgroups (tiles * s8 per XCD, 32 CUs, one workgroup per CU) quantise best against the fixed per-workgroup cost
  // (pipeline fill + the 256 KiB atomic epilogue, ~26 us measured, vs ~1.15 us per 32-token stage).
  int best = 1;
  double best_t = 1e30;
  for (int s8 = 1; s8 <= 32; ++s8) {
    const int per_try = (total_steps + 8 * s8 - 1) / (8 * s8);
    if (per_try < 8 && s8 > 1) break;
    const int units = tiles * s8;
    const double t = (double)((units + 31) / 32) * (26.0 + 1.15 * per_try);
    if (t < best_t * 0.98) { best_t = t; best = s8; }
  }
  int splits = 8 * best;
  if (const int forced = get_option(OPT_TN_SPLITS)) splits = forced;
  const int per = (total_steps + splits - 1) / splits;
  splits = (total_steps + per - 1) / per;
  const int slices8 = (splits + 7) / 8 * 8;  // grid covers a multiple of 8 slices (one per XCD per pass); empty ones exit
  const unsigned grid = (unsigned)(slices8 * tiles);
  const size_t lds = (size_t)NSTAGE * 2 * IMG_BYTES;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ALPRO_BF16) {
    static DeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(grid), dim3(NT3), lds, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, per, tn, tk, colsum);
  } else {
    static DeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL(gemm_tn_kernel<f16_t>, dim3(grid), dim3(NT3), lds, st, (const f16_t*)A, lda, (const f16_t*)B, ldb, C, ldc, M, N, K, per, tn, tk, colsum);
  }
  return check_launch("alpro_gemm_tn_acc");
}

extern "C" int alpro_colsum_acc(const void* A, int64_t lda, float* out, int dtype, int M, int N, void* stream) {
  ALPRO_CHECK(A && out && M > 0 && N > 0, "alpro_colsum_acc: bad args");
  const int esz = dtype == ALPRO_F32 ? 4 : 2;
  ALPRO_CHECK((N * esz) % 16 == 0 && (lda * esz) % 16 == 0, "alpro_colsum_acc: rows must be 16-byte multiples");
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(colsum_kernel<T>, dim3((N + 64 * Chunk<T>::N - 1) / (64 * Chunk<T>::N), (M + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const T*)A, lda, out, M, N));
  return check_launch("alpro_colsum_acc");
}
