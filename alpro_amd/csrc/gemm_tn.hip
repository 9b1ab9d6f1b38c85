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
//  * split over M: wgrad outputs are tiny (768..3072 x 768) while M is ~10^5, so the token dimension is cut into R ranges and a
//    workgroup takes one (range, tile) unit; R is chosen so that R x tiles fills the 256 CUs in one round (36 tiles -> 7 ranges);
//    the 1-D grid is decoded so that an XCD owns a contiguous run of units (tools/probe_xcd.hip) and the cross-XCD re-use of token
//    rows is left to the Infinity Cache; partial tiles are combined with hardware fp32 atomics (which is also what "accumulate
//    into .grad" needs) or, bit-reproducibly, through a workspace + tn_reduce_kernel (alpro_gemm_tn_acc_ws);
//  * optional bias gradient: the waves sum their dY fragments on the VALU (colsum), every wave an equal share.
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

template <typename T, int PP = 0>
__global__ __launch_bounds__(NT3, 1) void gemm_tn_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K, int per,
                                                         int tiles, int units, int tn_cnt, float* __restrict__ colsum, float* __restrict__ part,
                                                         float* __restrict__ part_cs, int ablate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // Work units: (token range, tile), range-major; a unit is `per` 32-token stages of one 256 x 256 tile.  The 1-D grid is decoded
  // so that XCD x (= blockIdx % 8; workgroups are dealt to the XCDs round-robin, tools/probe_xcd.hip) owns a CONTIGUOUS run of
  // units: the workgroups that share an L2 then work on the same token range and on tiles with a common dY / X column panel.  The
  // re-use across XCDs (every token row is wanted by all tiles of its range at about the same time) is served by the Infinity
  // Cache -- binding a range to one XCD, as earlier versions did, buys nothing measurable and costs the freedom to pick the
  // number of ranges that fills the 256 CUs in one round (profiles/r2_gemm_epilogue_experiments.txt, experiment 8).
  const int upx = (units + 7) >> 3;
  const int unit = (blockIdx.x & 7) * upx + (blockIdx.x >> 3);
  if (unit >= units) return;
  const int range = unit / tiles, tile = unit - range * tiles;
  const int total_steps = (M + TM - 1) / TM;
  const int s0 = range * per, s1 = min(s0 + per, total_steps);
  const int n0 = (tile % tn_cnt) * TW, k0 = (tile / tn_cnt) * TW;
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
  if constexpr (PP == 0) {
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
  } else {   // two-group schedule: stages s0, s0+1 complete and piece 0 of stage s0+2; the loop issues one piece pair per phase from there on
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int d = 0; d < 4; ++d) copy(d, p);
      advance(0);
      advance(1);
    }
    copy(0, 2);
    copy(1, 2);
    advance(0);
  }
  [[maybe_unused]] const int pos = wave >> 2;
  auto load_frags = [&](u32x4* fa, u32x4* fb, const char* cA, const char* cB, int ks) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = frag8(cB, ks, wc * 64 + j * 32, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag8(cA, ks, wr * 128 + i * 32, lane);
  };
  // Bias gradient for free: colsum[n] += sum over tokens of A[m, n].  The A fragments already sit in registers (lane = one
  // column, 8 tokens per fragment), so the waves add them up on the VALU while the MFMA pipe
  // runs -- the separate pass over dY that torch / alpro_colsum_acc would make never happens.  The four waves of a wave row hold
  // the SAME four A fragments: wave (wr, wc) sums fragment wc only, so the extra VALU work (16 converts + adds per 16 tokens) is
  // spread evenly -- with the wc == 0 waves doing all four, they arrived late at every barrier and the whole kernel ran 12 % slower.
  // ... and the k-tiles that share a dY column panel take turns: k-tile tk sums the stages with stage % (number of k-tiles) == tk,
  // so every wave of every workgroup carries the same small share and the lock-step of the waves is not disturbed (with the k0 == 0
  // workgroups doing all of it, a third of the workgroups of a K = 768 gradient finished 9 % late).
  const int tk_cnt = tiles / tn_cnt, tk = tile / tn_cnt;
  const bool do_cs = colsum != nullptr;
  int cs_turn = s0 % tk_cnt;
  float cs = 0.f;
  auto add_cols = [&](const u32x4* fa) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i == wc) {  // wave-uniform
        float f[8];
        unpack_chunk<T>(fa[i], f);
        cs += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
      }
  };
  if constexpr (PP == 0) {
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
      const bool cs_now = do_cs && cs_turn == tk;
      cs_turn = cs_turn + 1 == tk_cnt ? 0 : cs_turn + 1;
      if (cs_now) add_cols(fa0);
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
      if (cs_now) add_cols(fa1);
      advance(0);
    }
  } else {
    // Two-group ("ping-pong") schedule (round 4; the NT kernel's structure, gemm.hip gemm_nt256q_kernel, on this kernel's operands).  A stage
    // is two PHASES (token halves ks = 0 / 1): LOAD segment = the phase's six 8-token fragments (24 ds_read_b64_tr_b16) + one piece pair of a
    // later stage -> s_barrier -> MFMA segment = 8 MFMAs on 8 different accumulators at raised priority -> s_barrier.  The upper wave row
    // runs one barrier behind, so each SIMD always has one wave on the matrix pipe and one on LDS / copies.  Phase k = 2j + ks of stage j:
    //   copies: piece (k-1) & 1 of stage ((k-1) >> 1) + 3, i.e. phase 2j -> piece 1 of stage j+2, phase 2j+1 -> piece 0 of stage j+3, into the
    //           buffer last read in LOAD(2j-1) / LOAD(2j+1 - 2): two phases after its last reader (WAR, see gemm.hip);
    //   wait:   vmcnt(6) at the end of every LOAD segment: all but the last three phases' copies have landed, so the stage first read in
    //           the NEXT phase is complete before the barrier that separates the two (RAW).
    u32x4 fa[4], fb[2];
    auto pbar = [] {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // stage s0 landed (10 copies issued: stage s0+1 and piece 0 of s0+2 stay in flight)
    pbar();
    if (wr == 1) pbar();   // the upper wave row drops one barrier behind
    for (int st = s0; st < s1; ++st) {
      const int j = st - s0;
      const int cur = j & (NSTAGE - 1);
      const char* cA = smem + cur * 2 * IMG_BYTES;
      const bool cs_now = do_cs && cs_turn == tk;
      cs_turn = cs_turn + 1 == tk_cnt ? 0 : cs_turn + 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        load_frags(fa, fb, cA, cA + IMG_BYTES, ks);
        const int buf = (j + 2 + ks) & (NSTAGE - 1);
        copy(2 * (1 - ks), buf);          // phase 2j: piece 1 (A, B) of stage j+2; phase 2j+1: piece 0 (A, B) of stage j+3
        copy(2 * (1 - ks) + 1, buf);
        advance(1 - ks);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        pbar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) mma_chunk<T>(acc[i][jj], fa[i], fb[jj]);
        __builtin_amdgcn_s_setprio(0);
        if (cs_now) add_cols(fa);
        pbar();
      }
    }
    if (wr == 0) pbar();   // equal barrier counts for both wave rows
  }
  if (do_cs) {
    const float t = cs + __shfl_xor(cs, 32, 64);  // the two token halves of the fragment
    const int n = n0 + wr * 128 + wc * 32 + (lane & 31);
    if (lane < 32 && part_cs) part_cs[((int64_t)range * tk_cnt + tk) * (tn_cnt * TW) + n] = t;  // workspace mode: summed by tn_reduce_kernel
    else if (lane < 32 && n < N) unsafeAtomicAdd(colsum + n, t);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail copies (zero page) before the wave exits
#ifdef ALPRO_ABLATIONS
  if (ablate == 1 && acc[0][0][0] != 12345.f) return;  // measurement only (tn_kind 1, tools/ build): no epilogue, results are garbage
#endif
  if (part) {
    // Workspace mode: the partial tile goes out in accumulator order -- 16 bytes per lane, 1 KiB per wave instruction, 32 plain
    // stores per lane instead of 128 fabric atomics -- and tn_reduce_kernel adds the partials of a tile to C in a fixed order.
    float* dst = part + (int64_t)unit * (TW * TW) + wave * 8192 + lane * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          __builtin_nontemporal_store(v, (f32x4*)(dst + ((i * 2 + j) * 4 + q) * 256));
        }
    return;
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

// Second half of the workspace mode: C[n, k] += sum over token ranges of the partial tiles (fixed order: bit-reproducible), and the
// same for the bias-gradient partials.  One thread per 16-byte piece of a tile in accumulator order (see the store above): piece f
// of wave w = f >> 11 is rows wr*128 + i*32 + 8q + 4*(lane >> 5) + {0..3}, column wc*64 + j*32 + (lane & 31).
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_cs, float* __restrict__ C,
                                                        int64_t ldc, float* __restrict__ colsum, int N, int K, int tiles, int ranges, int tn_cnt) {
  const int tile_blocks = part ? tiles * 64 : 0;  // part == NULL: one token range, the tile went straight to C; only the bias partials are summed
  if ((int)blockIdx.x >= tile_blocks) {  // bias gradient: 32 columns per block, the partial sets dealt to 8 thread groups
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, p = threadIdx.x >> 5;
    const int n = (blockIdx.x - tile_blocks) * 32 + c;
    const int sets = ranges * (tiles / tn_cnt);  // one partial per (token range, k-tile)
    float t = 0.f;
    for (int r = p; r < sets; r += 8) t += part_cs[(int64_t)r * (tn_cnt * TW) + n];
    red[p][c] = t;
    __syncthreads();
    if (p == 0 && n < N) colsum[n] += ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
    return;
  }
  const int tile = blockIdx.x >> 6, f = ((blockIdx.x & 63) << 8) + threadIdx.x;
  const int lane = f & 63, q = (f >> 6) & 3, j = (f >> 8) & 1, i = (f >> 9) & 3, wave = f >> 11;
  const int n = (tile % tn_cnt) * TW + (wave >> 2) * 128 + i * 32 + 8 * q + 4 * (lane >> 5);
  const int k = (tile / tn_cnt) * TW + (wave & 3) * 64 + j * 32 + (lane & 31);
  const float* src = part + (int64_t)tile * (TW * TW) + f * 4;
  const int64_t stride = (int64_t)tiles * (TW * TW);
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  int r = 0;
  for (; r + 4 <= ranges; r += 4) {
    const f32x4 a = __builtin_nontemporal_load((const f32x4*)(src + (r + 0) * stride)), b = __builtin_nontemporal_load((const f32x4*)(src + (r + 1) * stride));
    const f32x4 c = __builtin_nontemporal_load((const f32x4*)(src + (r + 2) * stride)), d = __builtin_nontemporal_load((const f32x4*)(src + (r + 3) * stride));
    t = (((t + a) + b) + c) + d;
  }
  for (; r < ranges; ++r) t = t + __builtin_nontemporal_load((const f32x4*)(src + r * stride));
  if (k < K) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (n + c < N) C[(int64_t)(n + c) * ldc + k] += t[c];
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

namespace {
struct TnPlan {
  int tn, tiles, total_steps, ranges, per, units;
  size_t part_floats, cs_floats;
};
// Token ranges (see the kernel): R ranges x tiles workgroups, one workgroup per CU.  R is picked against a measured cost model --
// rounds over the 256 CUs x (fixed ~26 us per workgroup: pipeline fill + epilogue issue, + ~1.15 us per stage) plus what the R sets
// of partial tiles cost on the way to C: through fp32 fabric atomics ~2 TB/s (1.14 us per MB) when every workgroup finishes at
// once, through the workspace (plain stores + tn_reduce_kernel) about a third of that.  36 tiles -> 7 ranges = 252 workgroups
// in ONE round; 9 tiles -> 28 ranges; a 30522-row vocabulary projection (360 tiles) at 2560 tokens -> no split.
TnPlan tn_plan(int M, int N, int K, bool ws, int cus) {
  TnPlan p;
  p.tn = (N + TW - 1) / TW;
  p.tiles = p.tn * ((K + TW - 1) / TW);
  p.total_steps = (M + TM - 1) / TM;
  p.ranges = 1;
  const int forced = get_option(OPT_TN_SPLITS);
  if (forced > 0) {
    p.ranges = forced;
  } else {
    const double set_us = (ws ? 0.4e-6 : 1.14e-6) * 4.0 * (double)N * (double)K;
    double best_t = 1e30;
    // cus: 256, or fewer while a collective's kernels hold CUs (cu_budget of the launch stream): one workgroup per CU, so rounds are counted over these
    for (int r = 1; r <= 256; ++r) {
      const int per_try = (p.total_steps + r - 1) / r;
      if (per_try < 4 && r > 1) break;
      const double t = (double)(((long)p.tiles * r + cus - 1) / cus) * (26.0 + 1.15 * per_try) + set_us * r + (ws && r > 1 ? 6.0 : 0.0);
      if (t < best_t * 0.98) { best_t = t; p.ranges = r; }
    }
  }
  p.per = (p.total_steps + p.ranges - 1) / p.ranges;
  p.ranges = (p.total_steps + p.per - 1) / p.per;
  p.units = p.ranges * p.tiles;
  p.part_floats = p.ranges > 1 ? (size_t)p.units * TW * TW : 0;
  // bias-gradient partials: one per (token range, k-tile).  The workspace plan routes them through tn_reduce_kernel even for ONE range
  // (K = 3072 has 12 k-tiles): atomics from the k-tiles of a column panel would arrive in arbitrary order and the "bit-reproducible"
  // promise of the workspace path would not cover the bias gradient.
  p.cs_floats = (ws || p.ranges > 1) ? (size_t)p.ranges * (p.tiles / p.tn) * p.tn * TW : 0;
  return p;
}
}  // namespace

// The plan depends on the CU budget of the stream the launch will go to (round 5: per-stream option), which this query does not know:
// it answers with the largest workspace any budget's plan needs (25 plans; memoised per shape).
extern "C" size_t alpro_gemm_tn_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  struct Memo { int M, N, K, forced; size_t bytes; };
  static Memo memo[64];
  static int n_memo = 0, lock = 0;
  const int forced = get_option(OPT_TN_SPLITS);
  while (__atomic_exchange_n(&lock, 1, __ATOMIC_ACQUIRE)) {}
  for (int i = 0; i < n_memo; ++i)
    if (memo[i].M == M && memo[i].N == N && memo[i].K == K && memo[i].forced == forced) {
      const size_t b = memo[i].bytes;
      __atomic_store_n(&lock, 0, __ATOMIC_RELEASE);
      return b;
    }
  __atomic_store_n(&lock, 0, __ATOMIC_RELEASE);
  size_t best = 0;
  for (int cus = 64; cus <= 256; cus += 8) {
    const TnPlan p = tn_plan(M, N, K, true, cus);
    const size_t b = (p.part_floats + p.cs_floats) * sizeof(float);
    best = b > best ? b : best;
  }
  while (__atomic_exchange_n(&lock, 1, __ATOMIC_ACQUIRE)) {}
  if (n_memo < 64) memo[n_memo++] = Memo{M, N, K, forced, best};
  __atomic_store_n(&lock, 0, __ATOMIC_RELEASE);
  return best;
}

extern "C" int alpro_gemm_tn_ranges(int M, int N, int K, int compute_units) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int cus = (compute_units >= 64 && compute_units < 256) ? compute_units / 8 * 8 : 256;
  return tn_plan(M, N, K, true, cus).ranges;
}

extern "C" int alpro_gemm_tn_acc_ws(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int dtype, int M, int N,
                                    int K, float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
  ALPRO_CHECK(A && B && C && M > 0 && N > 0 && K > 0, "alpro_gemm_tn_acc: bad args");
  ALPRO_CHECK(dtype == ALPRO_BF16 || dtype == ALPRO_F16, "alpro_gemm_tn_acc: 16-bit operands only (fp32 mode uses alpro_transpose + alpro_gemm)");
  ALPRO_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= (N + 7) / 8 * 8 && ldb >= (K + 7) / 8 * 8,
              "alpro_gemm_tn_acc: lda/ldb must be multiples of 8 covering N/K rounded up to 8 (16-byte chunks are read whole)");
  ALPRO_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "alpro_gemm_tn_acc: operands must be 16-byte aligned");
  const bool ws = workspace != nullptr;
  const TnPlan p = tn_plan(M, N, K, ws, cu_budget((hipStream_t)stream));
  float* part = nullptr;
  float* part_cs = nullptr;
  if (ws) {
    ALPRO_CHECK(((uintptr_t)workspace % 16) == 0 && workspace_bytes >= (p.part_floats + p.cs_floats) * sizeof(float),
                "alpro_gemm_tn_acc_ws: workspace must be 16-byte aligned and hold alpro_gemm_tn_workspace_bytes(M, N, K) bytes");
    part = p.ranges > 1 ? (float*)workspace : nullptr;
    part_cs = colsum ? (float*)workspace + p.part_floats : nullptr;
  }
  const int kind_opt = get_option(OPT_TN_KIND);
  const bool pp = kind_opt == 2;             // 2: the two-group schedule (round 4), result-preserving
#ifdef ALPRO_ABLATIONS
  const int kind = kind_opt == 1 ? 1 : 0;    // 1: no epilogue (timing only; measurement build)
#else
  const int kind = 0;
#endif
  const unsigned grid = (unsigned)((p.units + 7) / 8 * 8);
  const size_t lds = (size_t)NSTAGE * 2 * IMG_BYTES;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ALPRO_BF16) {
    static DeviceOnce once;
    once.run([&] {
      (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (pp) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, 1>), dim3(grid), dim3(NT3), lds, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, p.per, p.tiles, p.units, p.tn, colsum, part, part_cs, kind);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, 0>), dim3(grid), dim3(NT3), lds, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, p.per, p.tiles, p.units, p.tn, colsum, part, part_cs, kind);
  } else {
    static DeviceOnce once;
    once.run([&] {
      (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<f16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<f16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (pp) hipLaunchKernelGGL((gemm_tn_kernel<f16_t, 1>), dim3(grid), dim3(NT3), lds, st, (const f16_t*)A, lda, (const f16_t*)B, ldb, C, ldc, M, N, K, p.per, p.tiles, p.units, p.tn, colsum, part, part_cs, kind);
    else hipLaunchKernelGGL((gemm_tn_kernel<f16_t, 0>), dim3(grid), dim3(NT3), lds, st, (const f16_t*)A, lda, (const f16_t*)B, ldb, C, ldc, M, N, K, p.per, p.tiles, p.units, p.tn, colsum, part, part_cs, kind);
  }
  if ((part || part_cs) && kind != 1)
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((part ? p.tiles * 64 : 0) + (part_cs ? p.tn * (TW / 32) : 0)), dim3(256), 0, st, part, part_cs, C, ldc, colsum, N, K, p.tiles, p.ranges, p.tn);
  return check_launch("alpro_gemm_tn_acc");
}

extern "C" int alpro_gemm_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int dtype, int M, int N,
                                 int K, float* colsum, void* stream) {
  return alpro_gemm_tn_acc_ws(A, lda, B, ldb, C, ldc, dtype, M, N, K, colsum, nullptr, 0, stream);
}

extern "C" int alpro_colsum_acc(const void* A, int64_t lda, float* out, int dtype, int M, int N, void* stream) {
  ALPRO_CHECK(A && out && M > 0 && N > 0, "alpro_colsum_acc: bad args");
  const int esz = dtype == ALPRO_F32 ? 4 : 2;
  ALPRO_CHECK((N * esz) % 16 == 0 && (lda * esz) % 16 == 0, "alpro_colsum_acc: rows must be 16-byte multiples");
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(colsum_kernel<T>, dim3((N + 64 * Chunk<T>::N - 1) / (64 * Chunk<T>::N), (M + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const T*)A, lda, out, M, N));
  return check_launch("alpro_colsum_acc");
}
