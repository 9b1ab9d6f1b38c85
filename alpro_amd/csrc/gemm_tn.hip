// Weight-gradient GEMM for gfx950:  C[N, K] += A[M, N]^T * B[M, K]   (A = dY, B = X, both row-major in a 16-bit
// dtype, contraction over the TOKEN dimension M, fp32 accumulation, fp32 atomic accumulate into C = param.grad).
//
// Both operands are read in their natural layout -- no transposed copies in HBM:
//  * 256 x 256 output tile, 8 waves (2 x 4, 128 x 64 each), contraction step = 64 tokens; LDS stage = two
//    (64 tokens x 256 columns) images filled by global_load_lds_dwordx4, double buffered (128 KiB);
//  * MFMA operands need 8 consecutive tokens of ONE column per lane: ds_read_b64_tr_b16 gathers 4 tokens x 16
//    columns per 16-lane group (semantics: tools/probe_tr.hip); the 16-byte chunk index is XOR-ed with
//    (token & 3) << 2 on the DMA source side so the four token rows of a gather sit in different 64-byte windows;
//  * split over M: wgrad outputs are tiny (768..3072 x 768) while M is ~10^5, so the token range is cut into
//    `splits` slices (grid.z) sized to fill the 256 CUs; partial tiles are combined with hardware fp32 atomics,
//    which is also what "accumulate into .grad" needs.
// Rows >= M / columns >= N,K read a zero page, so no operand needs padding.
#include "common.hpp"

namespace alpro {
namespace {

constexpr int TM = 64;          // tokens per stage
constexpr int TW = 256;         // tile width (columns of A -> rows of C; columns of B -> columns of C)
constexpr int ROW_BYTES = TW * 2;
constexpr int IMG_BYTES = TM * ROW_BYTES;  // 32 KiB
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
__global__ __launch_bounds__(NT3, 2) void gemm_tn_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K, int steps_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int n0 = blockIdx.x * TW, k0 = blockIdx.y * TW;
  const int total_steps = (M + TM - 1) / TM;
  const int s0 = blockIdx.z * steps_per_split;
  const int s1 = min(s0 + steps_per_split, total_steps);
  if (s0 >= s1) return;

  // DMA pieces: 1 KiB = 2 token rows x 512 B; wave w moves pieces w, w+8, w+16, w+24 of each image per stage.
  // LDS slot (row, chunk') receives source chunk chunk' ^ ((row & 3) << 2).
  int rowA[4], colA[4];  // token row within stage, source column (elements) within tile
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 8 * i) * 2 + (lane >> 5);
    const int ch = (lane & 31) ^ ((row & 3) << 2);
    rowA[i] = row;
    colA[i] = ch * 8;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  const char* zero = (const char*)g_zero_page;
  auto stage = [&](int step, int buf) {
    char* dA = smem + buf * 2 * IMG_BYTES;
    char* dB = dA + IMG_BYTES;
    const int64_t m_base = (int64_t)step * TM;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m_base + rowA[i];
      const bool mv = m < M;
      const char* sa = (mv && n0 + colA[i] < N) ? (const char*)(A + m * lda + n0 + colA[i]) : zero;
      const char* sb = (mv && k0 + colA[i] < K) ? (const char*)(B + m * ldb + k0 + colA[i]) : zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr)sa, (lds_ptr)(dA + (wave + 8 * i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr)sb, (lds_ptr)(dB + (wave + 8 * i) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  stage(s0, 0);
  __syncthreads();
  int cur = 0;
  for (int st = s0; st < s1; ++st) {
    if (st + 1 < s1) stage(st + 1, cur ^ 1);
    const char* cA = smem + cur * 2 * IMG_BYTES;
    const char* cB = cA + IMG_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 fa[4], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = frag8(cB, ks, wc * 64 + j * 32, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag8(cA, ks, wr * 128 + i * 32, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_chunk<T>(acc[i][j], fa[i], fb[j]);
    }
    __syncthreads();
    cur ^= 1;
  }
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
                                 int K, void* stream) {
  ALPRO_CHECK(A && B && C && M > 0 && N > 0 && K > 0, "alpro_gemm_tn_acc: bad args");
  ALPRO_CHECK(dtype == ALPRO_BF16 || dtype == ALPRO_F16, "alpro_gemm_tn_acc: 16-bit operands only (fp32 mode uses alpro_transpose + alpro_gemm)");
  ALPRO_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= (N + 7) / 8 * 8 && ldb >= (K + 7) / 8 * 8,
              "alpro_gemm_tn_acc: lda/ldb must be multiples of 8 covering N/K rounded up to 8 (16-byte chunks are read whole)");
  ALPRO_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "alpro_gemm_tn_acc: operands must be 16-byte aligned");
  const int tn = (N + TW - 1) / TW, tk = (K + TW - 1) / TW;
  const int total_steps = (M + TM - 1) / TM;
  // enough slices to put ~2 waves of workgroups on the 256 CUs, at least 8 stages per slice
  int splits = (512 + tn * tk - 1) / (tn * tk);
  if (splits > (total_steps + 7) / 8) splits = (total_steps + 7) / 8;
  if (splits < 1) splits = 1;
  const int per = (total_steps + splits - 1) / splits;
  splits = (total_steps + per - 1) / per;
  const size_t lds = 4 * IMG_BYTES;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ALPRO_BF16) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(tn, tk, splits), dim3(NT3), lds, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, per);
  } else {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    hipLaunchKernelGGL(gemm_tn_kernel<f16_t>, dim3(tn, tk, splits), dim3(NT3), lds, st, (const f16_t*)A, lda, (const f16_t*)B, ldb, C, ldc, M, N, K, per);
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
