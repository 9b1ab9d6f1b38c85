// Fused temporal half of a divided space-time block (round 6; vit.py:84-98,152-156 -- Attention.qkv, the (B*N, T, H, 64) softmax attention over
// the T frames of one patch, the head merge -- in ONE kernel): out = softmax_T((A Wq^T + bq)(A Wk^T + bk)^T * scale)(A Wv^T + bv), 16-bit
// operands, fp32 accumulation.  The (M, 2304) q | k | v tensor of the temporal branch never exists in HBM: at B = 32 that is 231 MB written by
// the qkv GEMM and read back by a separate attention launch (61 us of pure traffic) per block.
//
// One work tile = 256 token rows x ONE head: a 256 x 192 x K GEMM (q, k, v of that head) whose accumulators are consumed in place.
//  * 8 waves as 8 x 1: wave w owns token rows 32 w .. 32 w + 31 and ALL 192 columns (2 x 12 accumulator fragments of 16 x 16 = 96 registers),
//    so every (token group, head) is wave-private: no exchange between waves, no barrier in the epilogue.
//  * Orientation per fragment column: q and k are accumulated TRANSPOSED (v_mfma_f32_16x16x32 with the W fragment as the A operand: lane =
//    token, registers = 4 consecutive features), v normally (lane = feature, registers = 4 consecutive tokens).  A transposed q / k fragment
//    pair, packed to 16 bits, IS a K = 32 MFMA operand (lane = token row, 8 features per lane; the feature order is a permutation of d that
//    q and k share, and a dot product does not care), a normal v fragment IS (the lower half of) the A operand of the MFMA that contracts over tokens:
//        S^T[j][i] = sum_d k[j][d] q[i][d]     2 x v_mfma_f32_16x16x32     (lane = query i, registers = keys j = 4 (lane >> 4) + r)
//        O^T[d][i] = sum_j v[j][d] P^T[j][i]   4 x v_mfma_f32_16x16x32 (upper k half zero)   (lane = token i, registers = 4 consecutive d)
//    per 16-token fragment row (16 / T groups, block-diagonal mask), with the softmax on 4 registers per lane in between.  The attention of a
//    256 x 64 head tile costs 48 small MFMAs and ~350 VALU instructions per wave -- under 3 % of the tile's K loop.
//  * K loop: the 8-phase idea of gemm_nt256q_kernel (gemm.hip) on a 3-phase K-tile: phase p = the 64 W rows of part p (q, k, v) = 4 fragment
//    columns x 2 fragment rows x 2 k-steps = 16 MFMAs on 8 independent accumulators; LOAD segment (fragment reads + copies) -> barrier -> MFMA
//    segment -> barrier, waves 4-7 one barrier behind waves 0-3, so the two waves of a SIMD alternate on the matrix pipe.  Two K-tile parities
//    of {A: 8 wave-private 4 KiB row blocks, W0 / W1 / W2: 8 KiB each} = 112 KiB; refills (K-tile t, parity P; slots idle two phases after
//    their last read, as in the 8-phase kernel):
//        phase 0  reads A (own rows) + W0 (P)     copies W1 of K-tile t+1 -> parity P^1                                  MFMA q (transposed)
//        phase 1  reads W1 (P)                    copies W2 of K-tile t+1 -> P^1, own A rows (2 of 4 pieces) of t+2 -> P   MFMA k (transposed)
//        phase 2  reads W2 (P)                    copies W0 of K-tile t+2 -> P, own A rows (other 2 pieces) of t+2 -> P,
//                                                 s_waitcnt vmcnt(3): everything issued before this phase has landed        MFMA v
//    (A rows are read and refilled by the SAME wave: no cross-wave hazard; a W piece is copied by one wave and read by all: its wait sits
//    before the barrier that precedes the first read, one K-tile earlier.)  The K loop runs across tile boundaries like the 8-phase kernel's.
//  * W is addressed in place: part p of head h = rows p * H * 64 + h * 64 .. + 63 of the (3 H 64, K) qkv weight -- no permuted copy.
// Forward-only (inference): the training path keeps the separate launches, whose backward needs q, k, v.
#include <type_traits>

#include "common.hpp"

namespace alpro {
namespace {

constexpr int ROWB = 128;                  // bytes per LDS row: 64 16-bit elements of one K-tile
constexpr int TM = 256, NTH = 512;         // tile rows, threads
constexpr int A_BYTES = TM * ROWB;         // 32 KiB: 8 wave-private blocks of 32 rows
constexpr int WP_BYTES = 64 * ROWB;        // 8 KiB: the 64 W rows of one part
constexpr int PAR_BYTES = A_BYTES + 3 * WP_BYTES;   // 56 KiB per K-tile parity
constexpr int OST_BYTES = 8 * 4096;        // output staging: 32 tokens x 64 d x 2 B per wave
constexpr int LDS_BYTES = 2 * PAR_BYTES + OST_BYTES;

template <typename T> struct Mfma;
template <> struct Mfma<f16_t> {
  static __device__ __forceinline__ f32x4 k32(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma<bf16_t> {
  static __device__ __forceinline__ f32x4 k32(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

// Keep a value alive up to this point (no instruction).  Used behind the epilogue's small MFMAs: their A / B operands die with the instruction,
// and the register allocator then likes to give the result the SAME registers at a shifted offset (seen in the built object:
// `v_mfma ... v[58:61], v[60:61], v[70:71], 0`; LLVM only forbids the overlap for results wider than four registers).  With the operands
// live across the instruction there is nothing to overlap with -- a precaution, checked on the built object by tests/test_host_cpu.py.
template <typename X> __device__ __forceinline__ void keep_alive(const X& x) { asm volatile("" ::"v"(x)); }

struct TattnArgs {
  const void* A;        // (M, K) 16-bit, row stride lda elements: the normalised token rows (x[:, 1:] order: T consecutive rows = one patch)
  const void* W;        // (3 * H * 64, K) 16-bit, row stride ldw: Attention.qkv.weight
  const float* bias;    // (3 * H * 64) fp32 or nullptr
  void* out;            // (M, H * 64) 16-bit, row stride ldo: the attention output, heads merged (vit.py:96)
  void* qkv_out;        // training: (M, 3 * H * 64) 16-bit, row stride ldq -- q | k | v as the qkv Linear stores them (the backward reads them) -- or nullptr
  float* lse;           // training: log-sum-exp of every (32-row chunk, head, token), alpro_attn_temporal_fwd's layout ((chunk * H + h) * 32 + token), or nullptr
  int64_t lda, ldw, ldo, ldq;
  int M, H, K, Tn;
  float scale;
};

template <typename T>
__global__ __launch_bounds__(NTH, 2) void gemm_qkv_tattn_kernel(const TattnArgs g) {
  static_assert(sizeof(T) == 2, "16-bit operands only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                      // waves 4-7 run one barrier behind waves 0-3
  const int ntm = (g.M + TM - 1) / TM, nblk = ntm * g.H;
  const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  const int nk = g.K >> 6;                        // K-tiles of 64 elements: even, >= 2 (launcher)
  const uint32_t lds_base = lds_addr_of(smem);

  // tiles: row-panel-major (tile = tm * H + h), dealt to the XCDs in chunks of 32 like the 8-phase GEMM's static walk, so that the 12 heads
  // of an A row panel meet in one L2
  const uint32_t p = gridDim.x >> 3;
  auto list_tile = [&](int y, uint32_t j) -> int {
    const uint32_t t = ((j >> 5) << 8) + ((uint32_t)y << 5) + (j & 31u);
    return (j < 0x100000u && t < (uint32_t)nblk) ? (int)t : -1;
  };
  uint32_t walk = (uint32_t)(blockIdx.x >> 3);
  auto next_tile = [&]() -> int {
    const int t = list_tile((int)(blockIdx.x & 7u), walk);
    walk += p;
    return t;
  };

  // copy sources: lane -> (row lane >> 3 of an 8-row piece, 16-byte chunk (lane & 7) ^ swizzle(row)); everything tile-dependent is wave-uniform
  const int r8 = lane >> 3;
  struct Tile { const char* a; const char* w; };   // a: this wave's 32 rows of A (K-tile 0); w: row 0 of part 0 of the head (K-tile 0)
  auto tile_base = [&](int tile) {
    const int tm = tile / g.H, h = tile - tm * g.H;
    int row0 = tm * TM + wave * 32;
    if (row0 >= g.M) row0 = tm * TM;              // a wave without valid rows (ragged last panel: M % 32 == 0) re-reads the panel's first rows
    Tile t;
    t.a = (const char*)g.A + (int64_t)row0 * lda_b;
    t.w = (const char*)g.W + (int64_t)h * 64 * ldw_b;
    return t;
  };
  // piece i (0..3) of this wave's A rows of K-tile kt -> parity par
  auto copy_a = [&](const Tile& t, int kt, int i, int par) {
    const int row = i * 8 + r8;
    const uint32_t vo = (uint32_t)(row * lda_b) + (((uint32_t)(lane & 7) ^ (uint32_t)((row >> 1) & 7)) << 4);
    const char* kbase = t.a + (int64_t)kt * ROWB;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + par * PAR_BYTES + wave * 4096 + i * 1024);
    // reserved-register site (deliberate; -Werror=inline-asm otherwise): global_load_lds takes its LDS address from m0; listing it as clobbered keeps the compiler from assuming a value of its own survives the statement
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(kbase), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
  };
  // this wave's piece (rows 8 wave .. + 7) of W part `part` of K-tile kt -> parity par
  auto copy_w = [&](const Tile& t, int kt, int part, int par) {
    const int row = wave * 8 + r8;
    const uint32_t vo = (uint32_t)(row * ldw_b) + (((uint32_t)(lane & 7) ^ (uint32_t)((row >> 1) & 7)) << 4);
    const char* kbase = t.w + (int64_t)part * g.H * 64 * ldw_b + (int64_t)kt * ROWB;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + par * PAR_BYTES + A_BYTES + part * WP_BYTES + wave * 1024);
    // reserved-register site (deliberate; -Werror=inline-asm otherwise): see copy_a
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(kbase), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
  };

  // fragment read offsets: lane l supplies row (l & 15) and the 8-element k group (l >> 4) of a 16 x 32 operand fragment
  const int l15 = lane & 15, kg = lane >> 4;
  const int frag0 = l15 * ROWB + ((kg ^ ((l15 >> 1) & 7)) << 4);
  const char* aF[2] = {smem + wave * 4096 + frag0, smem + wave * 4096 + (frag0 ^ 64)};
  const char* wF[2] = {smem + A_BYTES + frag0, smem + A_BYTES + (frag0 ^ 64)};

  auto barrier = [] {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  int cur_t = next_tile();
  if (cur_t < 0) return;
  int nxt_t = next_tile();
  Tile cur = tile_base(cur_t);
  Tile nxt = tile_base(nxt_t >= 0 ? nxt_t : cur_t);
  // pipeline fill: K-tile 0 complete in parity 0; A and W0 of K-tile 1 in parity 1 (W1 / W2 of K-tile 1 go out in K-tile 0's phases 0 / 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) copy_a(cur, 0, i, 0);
#pragma unroll
  for (int part = 0; part < 3; ++part) copy_w(cur, 0, part, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) copy_a(cur, 1, i, 1);
  copy_w(cur, 1, 0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();

  const float sl = g.scale * 1.4426950408889634f;
  const int tsh = 31 - __builtin_clz((unsigned)g.Tn);   // Tn is a power of two <= 16 (launcher)
  while (true) {
    const int tile = cur_t;
    const int tm = tile / g.H, h = tile - tm * g.H;
    f32x4 acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (grp == 1) barrier();   // the upper group drops one barrier behind

    auto ktile = [&](auto par_tag, int t) __attribute__((always_inline)) {
      constexpr int P = decltype(par_tag)::value;
      const bool in1 = t + 1 < nk, in2 = t + 2 < nk;          // targets inside this tile? else the next tile's K-tile 0 / 1
      const int k1 = in1 ? t + 1 : 0, k2 = in2 ? t + 2 : t + 2 - nk;
      const Tile& t1 = in1 ? cur : nxt;
      const Tile& t2 = in2 ? cur : nxt;
      u32x4 fa[2][2], fw[4][2];
      auto load_a = [&]() {
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) fa[mf][ks] = *(const u32x4*)(aF[ks] + P * PAR_BYTES + mf * 16 * ROWB);
      };
      auto load_w = [&](int part) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) fw[nf][ks] = *(const u32x4*)(wF[ks] + P * PAR_BYTES + part * WP_BYTES + nf * 16 * ROWB);
      };
      auto mma = [&](int part, bool transposed) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
              acc[mf][part * 4 + nf] = transposed ? Mfma<T>::k32(fw[nf][ks], fa[mf][ks], acc[mf][part * 4 + nf])
                                                  : Mfma<T>::k32(fa[mf][ks], fw[nf][ks], acc[mf][part * 4 + nf]);
        __builtin_amdgcn_s_setprio(0);
      };
      // phase 0: q
      load_a();
      load_w(0);
      copy_w(t1, k1, 1, P ^ 1);
      barrier();
      mma(0, true);
      barrier();
      // phase 1: k
      load_w(1);
      copy_w(t1, k1, 2, P ^ 1);
      copy_a(t2, k2, 0, P);
      copy_a(t2, k2, 1, P);
      barrier();
      mma(1, true);
      barrier();
      // phase 2: v
      load_w(2);
      copy_w(t2, k2, 0, P);
      copy_a(t2, k2, 2, P);
      copy_a(t2, k2, 3, P);
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      barrier();
      mma(2, false);
      barrier();
    };
    for (int t = 0; t < nk; t += 2) {
      ktile(std::integral_constant<int, 0>{}, t);
      ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (grp == 0) barrier();   // re-align

    // ---- epilogue: the attention of this wave's 32 tokens x this head, out of the accumulators
    {
      int le = lane;
      asm volatile("" : "+v"(le));   // (an opaque copy of the lane id: everything lane-derived below is computed here, not hoisted over the K loop)
      const int el15 = le & 15, ekg = le >> 4;
      const int row0 = tm * TM + wave * 32;
      if (row0 < g.M) {
        float bq[4][4], bk[4][4], bv[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
          float c = 0.f;
          if (g.bias) {
            a = *(const f32x4*)(g.bias + h * 64 + nf * 16 + 4 * ekg);
            b = *(const f32x4*)(g.bias + (g.H + h) * 64 + nf * 16 + 4 * ekg);
            c = g.bias[(2 * g.H + h) * 64 + nf * 16 + el15];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { bq[nf][r] = a[r]; bk[nf][r] = b[r]; }
          bv[nf] = c;
        }
        char* ost = smem + 2 * PAR_BYTES + wave * 4096;
        // the staged 32 x 64 block (token rows of 128 bytes, 16-byte chunks XORed with the row) -> 16-byte row pieces at `dst` (row stride ld elements)
        auto drain = [&](T* dst, int64_t ld) __attribute__((always_inline)) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // DS operations of one wave complete in order; the staging area is wave-private
#pragma unroll
          for (int pz = 0; pz < 4; ++pz) {
            const int row = pz * 8 + (le >> 3), slot = le & 7;
            const u32x4 v = *(const u32x4*)(ost + row * 128 + ((slot ^ (row & 7)) << 4));
            store16_sc1(dst + (int64_t)row * ld + slot * 8, v);
          }
          asm volatile("" ::: "memory");
        };
        if (g.qkv_out) {   // training: q | k | v of this (32 tokens, head) to HBM as well, in the qkv Linear's layout
          T* qb = (T*)g.qkv_out + (int64_t)row0 * g.ldq + h * 64;
#pragma unroll
          for (int part = 0; part < 2; ++part) {   // q, k: lane = token, registers = 4 consecutive features -> 8-byte pieces
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
              for (int nf = 0; nf < 4; ++nf) {
                const f32x4 a = acc[mf][part * 4 + nf];
                const float* bb = part == 0 ? bq[nf] : bk[nf];
                const int row = mf * 16 + el15;
                *(u32x2*)(ost + row * 128 + (((2 * nf + (ekg >> 1)) ^ (row & 7)) << 4) + (ekg & 1) * 8) =
                    mk2(pack2(a[0] + bb[0], a[1] + bb[1], (T*)0), pack2(a[2] + bb[2], a[3] + bb[3], (T*)0));
              }
            drain(qb + (int64_t)part * g.H * 64, g.ldq);
          }
          // v: lane = feature, registers = 4 consecutive tokens.  Staged as 8-byte units (4 tokens x 1 feature) at [token group mf * 4 + kg][feature]; a
          // lane then reads the 8 units of (token group le >> 3, features 8 (le & 7) .. + 7) and permutes them into four 16-byte row pieces (the
          // packed epilogue of gemm_nt256q_kernel does the same)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
              const f32x4 a = acc[mf][8 + nf];
              *(u32x2*)(ost + ((mf * 4 + ekg) * 64 + nf * 16 + el15) * 8) = mk2(pack2(a[0] + bv[nf], a[1] + bv[nf], (T*)0), pack2(a[2] + bv[nf], a[3] + bv[nf], (T*)0));
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          {
            const int rg = le >> 3, cb = le & 7;
            u32x4 qv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) qv[j] = *(const u32x4*)(ost + (rg * 64 + cb * 8 + 2 * j) * 8);
            T* vb = qb + (int64_t)2 * g.H * 64 + (int64_t)(4 * rg) * g.ldq + cb * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              u32x4 o;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const uint32_t lo2 = (r & 2) ? qv[j].y : qv[j].x, hi2 = (r & 2) ? qv[j].w : qv[j].z;   // features 2 j / 2 j + 1, tokens (r & 2), (r & 2) + 1
                o[j] = __builtin_amdgcn_perm(hi2, lo2, (r & 1) ? 0x07060302u : 0x05040100u);
              }
              store16_sc1(vb + (int64_t)r * g.ldq, o);
            }
            asm volatile("" ::: "memory");
          }
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          // q^T, k^T fragment pairs -> K = 32 operands (lane = token, 8 features: registers of fragments 2 f and 2 f + 1)
          u32x4 qo[2], ko[2];
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            float qv[8], kv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int nf = 2 * f + (e >> 2), r = e & 3;
              qv[e] = acc[mf][nf][r] + bq[nf][r];
              kv[e] = acc[mf][4 + nf][r] + bk[nf][r];
            }
            qo[f] = pack_chunk<T>(qv);
            ko[f] = pack_chunk<T>(kv);
          }
          f32x4 st = {0.f, 0.f, 0.f, 0.f};
          st = Mfma<T>::k32(ko[0], qo[0], st);
          st = Mfma<T>::k32(ko[1], qo[1], st);    // S^T[j = 4 kg + r][i = l15]
          keep_alive(ko[0]); keep_alive(qo[0]); keep_alive(ko[1]); keep_alive(qo[1]);
          // softmax over the T keys of the query's own group, log2 domain
          float x[4], m = -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = ((4 * ekg + r) >> tsh) == (el15 >> tsh);
            x[r] = ok ? st[r] * sl : -INFINITY;
            m = fmaxf(m, x[r]);
          }
          m = fmaxf(m, __shfl_xor(m, 16, 64));
          m = fmaxf(m, __shfl_xor(m, 32, 64));
          float pr[4], sum = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pr[r] = __builtin_amdgcn_exp2f(x[r] - m);   // exp2(-inf) == 0 on the other groups' keys
            sum += pr[r];
          }
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          const float inv = 1.0f / sum;
          if (g.lse && ekg == 0) g.lse[((int64_t)(row0 >> 5) * g.H + h) * 32 + mf * 16 + el15] = (m + __builtin_amdgcn_logf(sum)) * 0.6931471805599453f;
          const u32x2 pt = mk2(pack2(pr[0], pr[1], (T*)0), pack2(pr[2], pr[3], (T*)0));   // P^T: B operand (lane = query, 4 keys)
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const u32x2 vo = mk2(pack2(acc[mf][8 + f][0] + bv[f], acc[mf][8 + f][1] + bv[f], (T*)0),
                                 pack2(acc[mf][8 + f][2] + bv[f], acc[mf][8 + f][3] + bv[f], (T*)0));   // v: A operand (lane = feature, 4 tokens)
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            // O^T[d = 16 f + 4 kg + r][i = l15].  The contraction over the fragment's 16 tokens runs on the K = 32 instruction with the upper four
            // k-elements of both operands zero: the K = 16 form (v_mfma_f32_16x16x16_f16 / _bf16) gave wrong values in two of four result
            // registers for rows 17 / 19 / 21 / 23 of every 32-token block on gfx950 -- in one build and not in the previous one, with or without
            // wait states behind it, with or without result / operand register overlap (round 6, profiles/r6_tattn_mfma_k16_finding.txt: K = 32 form 0 wrong of 589824,
            // every k16 variant ~8900); the K = 32 form is the one this chip's GEMMs run on
            const u32x4 va = mk4(vo.x, vo.y, 0u, 0u), pb = mk4(pt.x, pt.y, 0u, 0u);
            o = Mfma<T>::k32(va, pb, o);
            keep_alive(va); keep_alive(pb);
            const u32x2 ow = mk2(pack2(o[0] * inv, o[1] * inv, (T*)0), pack2(o[2] * inv, o[3] * inv, (T*)0));
            const int row = mf * 16 + el15;       // 16-byte chunk 2 f + (kg >> 1) of the token's 128-byte row, XORed with the row
            *(u32x2*)(ost + row * 128 + (((2 * f + (ekg >> 1)) ^ (row & 7)) << 4) + (ekg & 1) * 8) = ow;
          }
        }
        drain((T*)g.out + (int64_t)row0 * g.ldo + h * 64, g.ldo);
      }
    }
    if (nxt_t < 0) break;
    cur = nxt;
    cur_t = nxt_t;
    nxt_t = next_tile();
    nxt = tile_base(nxt_t >= 0 ? nxt_t : cur_t);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead copies went into dead slots: landed before the LDS belongs to someone else
}

}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_gemm_qkv_tattn(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int dtype,
                                    int M, int H, int T, int K, float scale, void* qkv_out, int64_t ldq, float* lse, void* stream) {
  ALPRO_CHECK(A && W && out && M > 0 && H > 0 && T > 0, "alpro_gemm_qkv_tattn: bad args");
  ALPRO_CHECK(dtype == ALPRO_BF16 || dtype == ALPRO_F16, "alpro_gemm_qkv_tattn: 16-bit operand dtypes only (the exact fp32 mode keeps the separate launches)");
  ALPRO_CHECK(16 % T == 0, "alpro_gemm_qkv_tattn: num_frm=%d must divide 16 (a frame group lives inside one 16-token accumulator fragment)", T);
  ALPRO_CHECK(M % 32 == 0, "alpro_gemm_qkv_tattn: M=%d must be a multiple of 32 rows (a wave owns 32 tokens)", M);
  ALPRO_CHECK(K % 128 == 0 && K >= 128, "alpro_gemm_qkv_tattn: K=%d must be a multiple of 128 (an even number of 64-deep K-tiles)", K);
  ALPRO_CHECK(((lda * 2) % 128) == 0 && ((ldw * 2) % 128) == 0 && (ldo % 8) == 0, "alpro_gemm_qkv_tattn: operand rows must be 128-byte aligned, output rows 16-byte aligned");
  ALPRO_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0), "alpro_gemm_qkv_tattn: pointers must be 16-byte aligned");
  ALPRO_CHECK((int64_t)M * lda * 2 < (int64_t)0x7FFF0000 && (int64_t)3 * H * 64 * ldw * 2 < (int64_t)0x7FFF0000, "alpro_gemm_qkv_tattn: operands beyond 2 GiB need 64-bit copy offsets");
  TattnArgs g;
  ALPRO_CHECK(!qkv_out || ((ldq % 8) == 0 && ((uintptr_t)qkv_out % 16) == 0), "alpro_gemm_qkv_tattn: qkv_out rows must be 16-byte aligned");
  g.A = A; g.W = W; g.bias = bias; g.out = out; g.qkv_out = qkv_out; g.lse = lse;
  g.lda = lda; g.ldw = ldw; g.ldo = ldo; g.ldq = ldq;
  g.M = M; g.H = H; g.K = K; g.Tn = T;
  g.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_qkv_tattn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_qkv_tattn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  const int grid = cu_budget(st);
  if (dtype == ALPRO_BF16) hipLaunchKernelGGL(gemm_qkv_tattn_kernel<bf16_t>, dim3(grid), dim3(NTH), LDS_BYTES, st, g);
  else hipLaunchKernelGGL(gemm_qkv_tattn_kernel<f16_t>, dim3(grid), dim3(NTH), LDS_BYTES, st, g);
  return check_launch("alpro_gemm_qkv_tattn");
}
