// Attention forward kernels for gfx950 (head_dim 64, wave64, MFMA 32x32).
//
// Both kernels compute, per 32-query tile and per wave,
//     S^T = K Q^T          (MFMA; lane l then owns query (l & 31) and 16 keys per 32-key tile, so the
//                            row softmax is lane-local plus ONE exchange with lane l ^ 32)
//     P   = softmax(S^T * scale + bias)      fp32 registers
//     O^T = V^T P^T        (MFMA; P is consumed straight from the accumulator registers as the
//                            B operand -- the MFMA k index is free to follow the C-layout key order)
// alpro_attn_fwd      : one workgroup per (sequence, head); K and V of that head resident in LDS
//                       (K XOR-swizzled for conflict-free ds_read_b128), 4 waves x 32-query tiles.
// alpro_attn_temporal : groups of T frames; 32 consecutive tokens (= 32/T groups) form one MFMA
//                       tile with a block-diagonal mask, one wave per (32 tokens, head).
// Storage dtype T: bf16/f16 (K=16 MFMA) or f32 (K=2 MFMA, exact mode); a 16-byte chunk is the unit.
#include "common.hpp"

namespace alpro {
namespace {

constexpr int HD = 64;

template <typename T> struct AttnCfg {
  static constexpr int E = sizeof(T);
  static constexpr int CN = 16 / E;        // elements per 16-byte chunk
  static constexpr int RB = HD * E;        // bytes per head row
  static constexpr int CPR = RB / 16;      // chunks per head row (8 or 16)
  static constexpr int KS = CPR / 2;       // MFMA chunk-steps over head_dim (two lane halves per step)
  static constexpr int CPT = 16 / CN;      // P chunks per 32-key tile (2 or 4)
};

template <typename T> __device__ __forceinline__ int k_swz(int row, int chunk) {
  if (AttnCfg<T>::CPR == 8) return chunk ^ ((row >> 1) & 7);  // 128-B rows: two rows per 256-B bank row
  return chunk ^ (row & 15);                                   // 256-B rows
}

// 4 consecutive keys of column d from a row-major (key, 64) LDS tile -> packed storage elements
template <typename T> struct Col4;
template <> struct Col4<float> {
  typedef u32x4 type;
  static __device__ __forceinline__ u32x4 load(const char* v, int key0, int d) {
    const float* p = (const float*)(v + key0 * AttnCfg<float>::RB) + d;
    return mk4(f2u(p[0]), f2u(p[64]), f2u(p[128]), f2u(p[192]));
  }
};
template <typename T16> struct Col4_16 {
  typedef u32x2 type;
  static __device__ __forceinline__ u32x2 load(const char* v, int key0, int d) {
    const uint16_t* p = (const uint16_t*)(v + key0 * (HD * 2)) + d;
    u32x2 r;
    r.x = (uint32_t)p[0] | ((uint32_t)p[64] << 16);
    r.y = (uint32_t)p[128] | ((uint32_t)p[192] << 16);
    return r;
  }
};
template <> struct Col4<bf16_t> : Col4_16<bf16_t> {};
template <> struct Col4<f16_t> : Col4_16<f16_t> {};

// V image in LDS: row-major (key, 64).  16-bit dtypes XOR the 16-byte chunk index with 4*bit1(key) so
// that the four key rows gathered by one ds_read_b64_tr_b16 land in distinct 64-B bank windows.
template <typename T> __device__ __forceinline__ int v_swz(int row, int chunk) {
  return sizeof(T) == 2 ? chunk ^ (((row >> 1) & 1) << 2) : chunk;
}

// A-operand chunk cc of the key tile at LDS `vt`: V[keys(cc, g)][d] with d = dt*32 + (lane & 31).
template <typename T> __device__ __forceinline__ u32x4 load_vt_chunk(const char* vt, int cc, int lane, int dt);
template <> __device__ __forceinline__ u32x4 load_vt_chunk<float>(const char* vt, int cc, int lane, int dt) {
  return Col4<float>::load(vt, 8 * cc + 4 * (lane >> 5), dt * 32 + (lane & 31));  // one quad per chunk: regs 4cc..4cc+3
}
// 16-bit: hardware transpose read.  ds_read_b64_tr_b16 semantics (probed on gfx950, tools/probe_tr.hip):
// within each 16-lane group, lane l receives element (l & 3) of the 8-byte piece addressed by lane
// (l >> 2) + 4j, j = 0..3.  Lane p therefore points at key row key0 + (p >> 2), d-quad (p & 3) of its
// group's 16-wide d block, and every lane gets 4 consecutive keys of its own d column.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 tr_quad(const char* vt, int key0, int lane, int dt) {
  const int p = lane & 15, seg = dt * 2 + ((lane >> 4) & 1);
  const int row = key0 + (p >> 2);
  const char* a = vt + row * 128 + ((seg ^ (((row >> 1) & 1) << 1)) << 5) + ((p & 3) << 3);
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
  return __builtin_bit_cast(u32x2, r);
}
template <typename T> __device__ __forceinline__ u32x4 load_vt_chunk16(const char* vt, int cc, int lane, int dt) {
  const int g = lane >> 5;
  const u32x2 a = tr_quad(vt, 16 * cc + 4 * g, lane, dt);      // regs 8cc..8cc+3
  const u32x2 b = tr_quad(vt, 16 * cc + 8 + 4 * g, lane, dt);  // regs 8cc+4..8cc+7
  const uint32_t ax = a.x, ay = a.y, bx = b.x, by = b.y;
  return mk4(ax, ay, bx, by);
}
template <> __device__ __forceinline__ u32x4 load_vt_chunk<bf16_t>(const char* vt, int cc, int lane, int dt) { return load_vt_chunk16<bf16_t>(vt, cc, lane, dt); }
template <> __device__ __forceinline__ u32x4 load_vt_chunk<f16_t>(const char* vt, int cc, int lane, int dt) { return load_vt_chunk16<f16_t>(vt, cc, lane, dt); }

// store 4 consecutive outputs d0..d0+3 of one query row
template <typename T> __device__ __forceinline__ void store_quad(T* dst, const float* v) {
  if constexpr (sizeof(T) == 4) {
    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    u32x2 u;
    u.x = pack2(v[0], v[1], (T*)0);
    u.y = pack2(v[2], v[3], (T*)0);
    *(u32x2*)dst = u;
  }
}

// softmax over NKT key tiles held in C layout.  `bias4(kt, rq)` returns the additive fp32 bias of the
// four keys of accumulator registers 4rq..4rq+3 (-inf masks a key).  p is normalised in place;
// returns the row's log-sum-exp.
template <int NKT, typename BiasF>
__device__ __forceinline__ float softmax_tiles(f32x16 (&s)[NKT], float scale, BiasF bias4) {
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float4 bq = bias4(kt, rq);
      const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = s[kt][4 * rq + e] * scale + bb[e];
        s[kt][4 * rq + e] = v;
        m = fmaxf(m, v);
      }
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(s[kt][r] - m);  // exp(-inf) == 0 for masked keys
      s[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] *= inv;
  return m + logf(sum);
}

// O^T (2 d-tiles) += V^T P^T over NKT key tiles; vt points at key 0 of the LDS V image
template <typename T, int NKT>
__device__ __forceinline__ void pv_tiles(f32x16 (&o)[2], const f32x16 (&p)[NKT], const char* vt, int lane) {
  typedef AttnCfg<T> C;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int cc = 0; cc < C::CPT; ++cc) {
      float pv[C::CN];
#pragma unroll
      for (int e = 0; e < C::CN; ++e) pv[e] = p[kt][cc * C::CN + e];
      const u32x4 b = pack_chunk<T>(pv);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const u32x4 a = load_vt_chunk<T>(vt + kt * 32 * C::RB, cc, lane, dt);
        mma_chunk<T>(o[dt], a, b);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the column reads of later key tiles from piling up in VGPRs
  }
}

// The same product with the V^T operand chunks fetched PDV steps ahead of the MFMA that consumes them (round 6).  pv_tiles() fences the
// scheduler per key tile (its reads would pile up in VGPRs otherwise), so every key tile began with an exposed LDS round trip, and inside a
// tile the reads were issued one step ahead: ~35 % of a query tile's cycles went to `s_waitcnt lgkmcnt` in front of an MFMA.  Step i =
// (key tile i / (2 CPT), P chunk (i / 2) % CPT, d half i & 1); the ring holds PDV chunks; a scheduling fence per step keeps the order.
#ifndef ATTN_PDV
#define ATTN_PDV 3   // ring depths of the 7-tile (ViT) form; measurement builds override them
#endif
#ifndef ATTN_PDK
#define ATTN_PDK 4
#endif
template <typename T, int NKT, int PDV = 3>
__device__ __forceinline__ void pv_tiles_pipelined(f32x16 (&o)[2], const f32x16 (&p)[NKT], const char* vt, int lane) {
  typedef AttnCfg<T> C;
  constexpr int NSTEP = NKT * C::CPT * 2;
  u32x4 ring[PDV];
  auto vchunk = [&](int i) __attribute__((always_inline)) {
    const int kt = i / (2 * C::CPT), cc = (i >> 1) % C::CPT, dt = i & 1;
    return load_vt_chunk<T>(vt + kt * 32 * C::RB, cc, lane, dt);
  };
#pragma unroll
  for (int i = 0; i < PDV && i < NSTEP; ++i) ring[i] = vchunk(i);
  u32x4 b = mk4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int i = 0; i < NSTEP; ++i) {
    const int kt = i / (2 * C::CPT), cc = (i >> 1) % C::CPT, dt = i & 1;
    if (dt == 0) {
      float pv[C::CN];
#pragma unroll
      for (int e = 0; e < C::CN; ++e) pv[e] = p[kt][cc * C::CN + e];
      b = pack_chunk<T>(pv);
    }
    mma_chunk<T>(o[dt], ring[i % PDV], b);
    if (i + PDV < NSTEP) ring[i % PDV] = vchunk(i + PDV);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T>
__device__ __forceinline__ void store_o(T* out_row, const f32x16 (&o)[2], int lane) {
  const int g = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float v[4] = {o[dt][4 * rq], o[dt][4 * rq + 1], o[dt][4 * rq + 2], o[dt][4 * rq + 3]};
      store_quad<T>(out_row + dt * 32 + 8 * rq + 4 * g, v);
    }
}

// ================================================================================================
// full attention: grid = batch * H workgroups of 256 threads
template <typename T, int NKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int H, float scale,
                                                       const float* __restrict__ key_bias, float* __restrict__ lse, float drop_p,
                                                       uint32_t drop_seed) {
  typedef AttnCfg<T> C;
  constexpr int LP = NKT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + LP * C::RB;
  float* Bs = (float*)(smem + 2 * LP * C::RB);  // additive key bias; -inf on the padded keys
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  for (int c = tid; c < LP; c += 256) Bs[c] = c < L ? (key_bias ? key_bias[(int64_t)b * L + c] : 0.f) : -INFINITY;
  const int64_t ldq = 3 * (int64_t)H * HD;  // elements per token row of qkv
  const T* base = qkv + (int64_t)b * L * ldq + h * HD;

  // stage K (swizzled) and V (row-major), zero-filling the padded keys
  for (int c = tid; c < LP * C::CPR; c += 256) {
    const int row = c / C::CPR, ch = c - row * C::CPR;
    u32x4 kv = mk4(0, 0, 0, 0), vv = mk4(0, 0, 0, 0);
    if (row < L) {
      const T* src = base + (int64_t)row * ldq + ch * C::CN;
      kv = *(const u32x4*)(src + H * HD);
      vv = *(const u32x4*)(src + 2 * H * HD);
    }
    *(u32x4*)(Ks + row * C::RB + (k_swz<T>(row, ch) << 4)) = kv;
    *(u32x4*)(Vs + row * C::RB + (v_swz<T>(row, ch) << 4)) = vv;
  }
  __syncthreads();

  const int nqt = (L + 31) >> 5;
  const int g = lane >> 5, ql = lane & 31;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q = qt * 32 + ql;
    const int qc = min(q, L - 1);
    u32x4 qf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[ks] = *(const u32x4*)(base + (int64_t)qc * ldq + (2 * ks + g) * C::CN);

    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      const int krow = kt * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const u32x4 a = *(const u32x4*)(Ks + krow * C::RB + (k_swz<T>(krow, 2 * ks + g) << 4));
        mma_chunk<T>(s[kt], a, qf[ks]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float l_se = softmax_tiles<NKT>(s, scale, [&](int kt, int rq) { return *(const float4*)(Bs + kt * 32 + 8 * rq + 4 * g); });
    if (lse && g == 0 && q < L) lse[((int64_t)b * H + h) * L + q] = l_se;
    if (drop_seed) {  // attention-probability dropout (xbert.py:331): mask is a pure function of (b, h, q, key)
      const uint32_t th = drop_thresh24(drop_p);
      const float ks = 1.0f / (1.0f - drop_p);
      const uint64_t base_i = (((uint64_t)b * H + h) * L + (uint64_t)qc) * L;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = drop_keep(drop_seed, base_i + kt * 32 + acc_row(r, lane), th) ? s[kt][r] * ks : 0.f;
    }

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    pv_tiles<T, NKT>(o, s, Vs, lane);
    if (q < L) store_o<T>(out + ((int64_t)b * L + q) * H * HD + h * HD, o, lane);
  }
}

// ================================================================================================
// full attention, 16-bit storage, throughput form.  Same math as attn_fwd_kernel; what changes is the schedule:
//  * <= 256 registers and ~75 KiB of LDS per workgroup, so TWO workgroups (8 waves, 2 per SIMD) share a CU and one
//    workgroup's K/V staging and softmax VALU work overlap the other's MFMAs (the fp32-capable kernel above needs
//    350 registers at 7 key tiles = one wave per SIMD, every phase exposed);
//  * K and V go global -> LDS by DMA (no VGPR round trip); the bank swizzle is applied on the source side;
//  * softmax in the log2 domain: p = exp2(s * scale*log2(e) - m) is one FMA + one v_exp_f32 per score on fully valid
//    key tiles, the row maximum is taken on the raw scores, and the 1/sum normalisation is applied to the 32x64
//    output tile instead of the 32xL probabilities;
//  * the next query tile's Q fragments are prefetched under the current tile's softmax;
//  * O^T is transposed through 4 KiB of wave-private LDS so that a row's 128 bytes leave in 16-byte stores.
__device__ u32x4 g_attn_zero[4];
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// The precise CLS query (round 4: alpro_amd.config.cls_precise; stand-alone form: cls_precise.hip alpro_attn_cls_fwd) -- round 6 form.  The
// CLS token's fp32 q row is carried THROUGH the 16-bit MFMA path as NS extra query columns whose values add up to it: x = p0 + p1 / C1
// (+ p2 / (C1 C2)), every part exactly representable in T (fp16: 11 + 11 bits, the low part scaled by 2^11 into the normal range; bf16:
// 8 + 8 + 8 bits).  A product of two 16-bit values is exact in the MFMA's fp32 accumulator, so the sum of the parts' score columns IS the
// fp32 score of the unrounded q (to 2^-22), and the same split of the unnormalised probabilities carries P into P V without its 16-bit
// rounding.  The columns are padded query slots of the last query tile (L = 197: 27 of them; an extra query tile when fewer than NS are
// free), so the side path costs a few lane exchanges in ONE wave instead of rounds 4-5's VALU pass over all of K and V by all 256 threads
// (+31 us on a 166 us launch at B = 64) -- and the 8-key-tile instantiation no longer needs a second register budget.
template <typename T> struct ClsSplit;
template <> struct ClsSplit<f16_t> { static constexpr int NS = 2; static constexpr float C1 = 2048.f, C2 = 1.f; };
template <> struct ClsSplit<bf16_t> { static constexpr int NS = 3; static constexpr float C1 = 1.f, C2 = 1.f; };
template <typename T> __device__ __forceinline__ void cls_split(float x, float& p0, float& p1, float& p2) {
  p0 = quantize<T>(x);
  const float r1 = (x - p0) * ClsSplit<T>::C1;   // exact: the residual of a rounding is representable, the scale is a power of two
  p1 = quantize<T>(r1);
  p2 = ClsSplit<T>::NS == 3 ? quantize<T>((r1 - p1) * ClsSplit<T>::C2) : 0.f;
}
// The same decomposition by TRUNCATION (round 6): the leading part keeps the operand dtype's explicit mantissa bits of x (one AND; exactly
// representable in the dtype for normal values), the residual of a truncation is exact like that of a rounding, and the LAST part is rounded by
// the 16-bit pack that follows anyway: 10 + 11 (fp16) / 7 + 7 + 8 (bf16) mantissa bits, i.e. x to 2^-21 / 2^-22 -- for 3 instead of 8-10 VALU
// instructions per value (the cvt pairs of quantize() were two thirds of the ~1900 instructions the CLS parts cost their wave).
template <typename T> struct ClsTrunc;
template <> struct ClsTrunc<f16_t> { static constexpr uint32_t MASK = 0xFFFFE000u; };
template <> struct ClsTrunc<bf16_t> { static constexpr uint32_t MASK = 0xFFFF0000u; };
__device__ __forceinline__ float and_bits(float x, uint32_t m) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & m); }

template <typename T> __device__ __forceinline__ float cls_join(float v0, float v1, float v2) {
  float r = fmaf(v1, 1.0f / ClsSplit<T>::C1, v0);
  if (ClsSplit<T>::NS == 3) r = fmaf(v2, 1.0f / (ClsSplit<T>::C1 * ClsSplit<T>::C2), r);
  return r;
}

// lane i <- lane i + N / lane i - N inside its row of 16 lanes: one VALU instruction (v_mov_b32_dpp row_shl / row_shr), where __shfl_down / __shfl_up
// go through the LDS crossbar (ds_bpermute + a wait each: the 250 exchanges of a CLS tile cost that wave ~12 us that way, measured in round 6)
template <int N> __device__ __forceinline__ float dpp_from_above(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
template <int N> __device__ __forceinline__ float dpp_from_below(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}

template <typename T, int NKT, bool HAS_BIAS, bool CLS = false, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_fwd16_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int H, float scale,
                                                            const float* __restrict__ key_bias, float* __restrict__ lse, float drop_p,
                                                            uint32_t drop_seed, int order, const float* __restrict__ cls_q, int cls_group,
                                                            float* __restrict__ cls_out) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int LP = NKT * 32, RB = 128;
  constexpr int NS = ClsSplit<T>::NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + LP * RB;
  char* Os = smem + 2 * LP * RB;
  constexpr bool HALF = NKT == 8;         // 8 key tiles: 2 KiB of output staging per wave keeps two workgroups per CU
  constexpr int OW = HALF ? 2048 : 4096;
  float* Bs = (float*)(Os + 4 * OW);      // additive key bias * log2(e); -inf on the padded keys
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, h;
  attn_unit(blockIdx.x, gridDim.x, H, order, b, h);
  for (int c = tid; c < LP; c += 256) Bs[c] = c < L ? (HAS_BIAS ? key_bias[(int64_t)b * L + c] * LOG2E : 0.f) : -INFINITY;
  const int64_t ldq = 3 * (int64_t)H * HD;  // elements per token row of qkv
  const T* base = qkv + (int64_t)b * L * ldq + h * HD;
  const int nqt = (L + 31) >> 5;
  const int g = lane >> 5, ql = lane & 31;

  // where the CLS query's NS parts ride: free query slots c0 .. c0 + NS - 1 of the last tile (inside one 16-lane row), else of an extra tile
  int cls_tile = -1, c0 = 0, ntile = nqt;
  if (CLS && cls_q) {   // (cls_q == nullptr with CLS compiled in: the dropout launches without a precise CLS query, see launch_attn16)
    c0 = L - 32 * (nqt - 1);
    if ((c0 & 15) + NS > 16) c0 = (c0 | 15) + 1;
    cls_tile = nqt - 1;
    if (c0 + NS > 32) { c0 = 0; cls_tile = nqt; ntile = nqt + 1; }
  }

  // K (chunk ^ ((row >> 1) & 7)) and V (chunk ^ 4*bit1(row)) images: 1 KiB pieces of 8 rows; padded rows <- zero page.  All of K first,
  // then this wave's first Q fragments, then V: the first query tile's scores and softmax run while V is still on its way (round 6).
  const uint32_t k_lds = lds_addr_of(Ks), v_lds = lds_addr_of(Vs);
  const char* zero = (const char*)g_attn_zero;
#pragma unroll
  for (int i = 0; i < NKT; ++i) {
    const int piece = wave + 4 * i;
    const int row = piece * 8 + (lane >> 3), slot = lane & 7;
    const T* src = base + (int64_t)row * ldq;
    dma16(row < L ? (const char*)(src + H * HD + ((slot ^ ((row >> 1) & 7)) << 3)) : zero, __builtin_amdgcn_readfirstlane(k_lds + piece * 1024));
  }
  auto load_q = [&](int qt, u32x4(&qf)[4]) __attribute__((always_inline)) {
    const int qc = min(qt * 32 + ql, L - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const u32x4*)(base + (int64_t)qc * ldq + (2 * ks + g) * 8);
    if constexpr (CLS) {
      if (qt == cls_tile) {   // (wave-uniform) lanes c0 + j carry part j of the fp32 q row of this sequence's CLS token
        const float* cq = cls_q + (int64_t)(b / cls_group) * ldq + h * HD;
        const int j = ql - c0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float4 a = *(const float4*)(cq + (2 * ks + g) * 8), c = *(const float4*)(cq + (2 * ks + g) * 8 + 4);
          const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
          float part[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {   // truncation split (ClsTrunc); the pack below rounds the last part
            const float p0 = and_bits(x[e], ClsTrunc<T>::MASK);
            const float r1 = (x[e] - p0) * ClsSplit<T>::C1;
            float p1 = r1, p2 = 0.f;
            if (NS == 3) { p1 = and_bits(r1, ClsTrunc<T>::MASK); p2 = (r1 - p1) * ClsSplit<T>::C2; }
            part[e] = j == 0 ? p0 : (j == 1 ? p1 : p2);
          }
          const u32x4 pk = pack_chunk<T>(part);
          if (j >= 0 && j < NS) qf[ks] = pk;
        }
      }
    }
  };
  // query tiles of this wave: t, t + 4, ... with t rotated by the workgroup index, so that the wave with one tile less (7 tiles on 4 waves)
  // is not the same SIMD's in the two workgroups that share a CU
#ifdef ATTN_NO_ROT
  int seq = wave;
#else
  int seq = (wave + (int)(blockIdx.x >> 3)) & 3;
#endif
  // The tile that carries the CLS parts costs its wave ~2000 more VALU instructions (part splits, DPP joins over all NKT * 16 score registers,
  // twice).  With 7 tiles on 4 waves the walk above gives it -- the last tile -- to a wave that has two tiles, i.e. puts the extras on the
  // workgroup's critical path; swapped with the one tile of the wave that has a round less (positions 3 <-> 6 at L = 197) they ride for free.
  int swap_a = -1, swap_b = -1;
#ifdef ATTN_NO_CLS_SWAP   // (measurement builds, tools/build_attn_variants.sh)
  if (false) {
#else
  if (CLS && cls_tile >= 0 && cls_tile == ntile - 1 && (ntile & 3) != 0 && (cls_tile & 3) != 3 && ntile > 4) {
#endif
    swap_a = cls_tile;
    swap_b = ((ntile - 1) & ~3) - 1;   // the last position of the walk that starts at 3: the wave with one tile less
  }
  auto tile_at = [&](int pos) __attribute__((always_inline)) { return pos == swap_a ? swap_b : (pos == swap_b ? swap_a : pos); };
  int qt = tile_at(seq);
  bool has = seq < ntile;
  u32x4 qf[4];
  load_q(has ? qt : 0, qf);
#pragma unroll
  for (int i = 0; i < NKT; ++i) {
    const int piece = wave + 4 * i;
    const int row = piece * 8 + (lane >> 3), slot = lane & 7;
    const T* src = base + (int64_t)row * ldq;
    dma16(row < L ? (const char*)(src + 2 * H * HD + ((slot ^ (((row >> 1) & 1) << 2)) << 3)) : zero, __builtin_amdgcn_readfirstlane(v_lds + piece * 1024));
  }
#ifdef ATTN_NO_KV_SPLIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKT) : "memory");   // K and Q have landed (in-order return); the NKT V pieces may still be in flight
#endif
  __syncthreads();

  const float sl = scale * LOG2E;
  typedef float f32x2v_t __attribute__((ext_vector_type(2)));
  char* Ow = Os + wave * OW;
  f32x16 s[NKT];
  float inv = 0.f;
  u32x4 qn[4];

  // ---- S^T = K Q^T and the softmax (log2 domain) of query tile qt; leaves the unnormalised probabilities in s and 1 / sum in inv
  auto scores = [&]() __attribute__((always_inline)) {
    const int q = qt * 32 + ql;
    // K fragments PDK steps ahead of their MFMA (round 6; step i = (k-step i / NKT, key tile i % NKT): consecutive MFMAs hit different
    // accumulators).  The compiler's own order kept two reads in flight and waited for one in front of every MFMA.  ViT shape -3 % on one box
    // (profiles/r6_attn_fwd_rings.txt); with 8 key tiles the 128 score registers leave no room for the ring (spills, +2..4 %): compiler order.
    if constexpr (NKT <= 7) {
      constexpr int NF = NKT * 4, PDK = ATTN_PDK;
      auto kfrag = [&](int i) __attribute__((always_inline)) {
        const int ks = i / NKT, krow = (i % NKT) * 32 + ql;
        return *(const u32x4*)(Ks + krow * RB + (((2 * ks + g) ^ ((krow >> 1) & 7)) << 4));
      };
      u32x4 ka[PDK];
#pragma unroll
      for (int i = 0; i < PDK; ++i) ka[i] = kfrag(i);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int ks = i / NKT, kt = i % NKT;
        if (ks == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          s[kt] = z;
        }
        mma_chunk<T>(s[kt], ka[i % PDK], qf[ks]);
        if (i + PDK < NF) ka[i % PDK] = kfrag(i + PDK);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int krow = kt * 32 + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 a = *(const u32x4*)(Ks + krow * RB + (((2 * ks + g) ^ ((krow >> 1) & 7)) << 4));
          if (ks == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            s[kt] = z;
          }
          mma_chunk<T>(s[kt], a, qf[ks]);
        }
      }
    }
    load_q(tile_at(min(seq + 4, ntile - 1)), qn);  // lands under the softmax
    const bool cls_here = CLS && qt == cls_tile;
    if constexpr (CLS) {
      if (cls_here) {   // lane c0: the fp32 score of the unrounded CLS query = the sum of its parts' columns.  The weights are per-lane constants
        // (zero outside lane c0: raw scores are finite, so 0 * neighbour adds nothing): one DPP + one FMA per part and register, no select
        const float w1 = ql == c0 ? 1.0f / ClsSplit<T>::C1 : 0.f, w2 = ql == c0 ? 1.0f / (ClsSplit<T>::C1 * ClsSplit<T>::C2) : 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float a = s[kt][r];
            float v = fmaf(dpp_from_above<1>(a), w1, a);
            if (NS == 3) v = fmaf(dpp_from_above<2>(a), w2, v);
            s[kt][r] = v;
          }
      }
    }
    // tiles with all 32 keys valid and no bias skip the bias FMA; quads of 4 keys that lie entirely beyond L are not evaluated at all
    float m = -INFINITY;
    typedef __attribute__((address_space(3))) const char lds_cchar_t;
    uint32_t bs_lane = 16u * (uint32_t)g;   // this lane's 4 of a quad's 8 keys; opaque per tile: hoisted over the tile loop, the 32 bias addresses of the
    asm volatile("" : "+v"(bs_lane));       // 8-tile instantiations were kept in (spilled) registers instead of one base + immediate offsets
    lds_cchar_t* bs_l = (lds_cchar_t*)(__attribute__((address_space(3))) char*)Bs + bs_lane;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (!HAS_BIAS && (kt + 1) * 32 <= L) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(m, s[kt][r]), s[kt][r + 1]);
      }
    }
    m *= sl;  // scale > 0
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (HAS_BIAS || (kt + 1) * 32 > L) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          if (kt < NKT - 1 || kt * 32 + 8 * rq < L) {   // (wave-uniform; only the LAST key tile is looked at quad by quad: 16 more basic blocks per tile otherwise)
            const f32x4 bq = *(const __attribute__((address_space(3))) f32x4*)(bs_l + (kt * 32 + 8 * rq) * 4);
            const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = fmaf(s[kt][4 * rq + e], sl, bb[e]);
              s[kt][4 * rq + e] = v;
              m = fmaxf(m, v);
            }
          }
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const f32x2v_t sl2 = {sl, sl}, nm2 = {-m, -m};
    f32x2v_t sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (!HAS_BIAS && (kt + 1) * 32 <= L) {
#pragma unroll
#ifdef ATTN_NO_PK
        for (int r = 0; r < 16; ++r) {
          const float pr = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sl, -m));
          s[kt][r] = pr;
          sum2.x += pr;
        }
#else
        for (int r = 0; r < 16; r += 2) {
          const f32x2v_t x = (f32x2v_t){s[kt][r], s[kt][r + 1]} * sl2 + nm2;   // v_pk_fma_f32
          const f32x2v_t pr = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
          s[kt][r] = pr.x;
          s[kt][r + 1] = pr.y;
          sum2 += pr;
        }
#endif
      } else {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          if (kt < NKT - 1 || kt * 32 + 8 * rq < L) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const f32x2v_t x = (f32x2v_t){s[kt][4 * rq + e], s[kt][4 * rq + e + 1]} + nm2;
              const f32x2v_t pr = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};  // exp2(-inf) == 0 for masked keys
              s[kt][4 * rq + e] = pr.x;
              s[kt][4 * rq + e + 1] = pr.y;
              sum2 += pr;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kt][4 * rq + e] = 0.f;
          }
        }
      }
    }
    float sum = sum2.x + sum2.y;
    sum += __shfl_xor(sum, 32, 64);
    inv = 1.0f / sum;
    if (lse && g == 0 && q < L) lse[((int64_t)b * H + h) * L + q] = (m + __builtin_amdgcn_logf(sum)) * LN2;
    if constexpr (DROP) {  // attention-probability dropout (xbert.py:331): mask is a pure function of (b, h, q, key); the CLS parts take query 0's.  (A template
      // parameter since round 6: as a runtime branch its 112 hash indices were hoisted over the tile loop and spilled in every instantiation)
      const uint32_t th = drop_thresh24(drop_p);
      const float ks = 1.0f / (1.0f - drop_p);
      int qd = (cls_here && ql >= c0 && ql < c0 + NS) ? 0 : min(q, L - 1);
      asm volatile("" : "+v"(qd));
      int lo = lane;
      asm volatile("" : "+v"(lo));   // an opaque copy of the lane id, per tile: hoisted over the tile loop, the 112 per-register index terms were spilled
      const uint64_t base_i = (((uint64_t)b * H + h) * L + (uint64_t)qd) * L + (uint64_t)(4 * (lo >> 5));
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = drop_keep(drop_seed, base_i + (uint64_t)(kt * 32 + acc_row(r, 0)), th) ? s[kt][r] * ks : 0.f;
    }
    if constexpr (CLS) {
      if (cls_here) {   // lane c0's probabilities, unrounded, as NS columns of 16-bit values that add up to them (truncation split, ClsTrunc: the
        // mask is all ones outside lane c0, so the other lanes keep their value and their residual is an exact zero; the pack in pv_tiles
        // rounds the last part)
        // (the lane selects are bit selects on per-lane masks -- v_bfi_b32: written as ?: the compiler turned each of the 112 into an
        // exec-masked branch, ~8 scalar instructions and two jumps per register)
        const uint32_t tm = ql == c0 ? ClsTrunc<T>::MASK : 0xFFFFFFFFu;
        const uint32_t sel1 = ql == c0 + 1 ? 0xFFFFFFFFu : 0u, sel2 = (NS == 3 && ql == c0 + 2) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pr = s[kt][r];
            const uint32_t hb = __builtin_bit_cast(uint32_t, pr) & tm;
            const float h0 = __builtin_bit_cast(float, hb);
            const float r1 = (pr - h0) * ClsSplit<T>::C1;
            uint32_t o;
            if (NS == 2) {
              const uint32_t d1 = __builtin_bit_cast(uint32_t, dpp_from_below<1>(r1));
              o = (d1 & sel1) | (hb & ~sel1);
            } else {
              const float h1 = and_bits(r1, ClsTrunc<T>::MASK);
              const float r2 = (r1 - h1) * ClsSplit<T>::C2;
              const uint32_t d1 = __builtin_bit_cast(uint32_t, dpp_from_below<1>(h1)), d2 = __builtin_bit_cast(uint32_t, dpp_from_below<2>(r2));
              o = (d1 & sel1) | (hb & ~sel1);
              o = (d2 & sel2) | (o & ~sel2);
            }
            s[kt][r] = __builtin_bit_cast(float, o);
          }
      }
    }
  };

  // ---- O^T = V^T P^T of query tile qt, normalised, out through wave-private LDS as 16-byte row pieces
  auto output = [&]() __attribute__((always_inline)) {
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#ifdef ATTN_NO_PIPE_V   // (measurement builds, tools/build_attn_variants.sh)
    pv_tiles<T, NKT>(o, s, Vs, lane);
#else
    if constexpr (NKT <= 7) pv_tiles_pipelined<T, NKT, ATTN_PDV>(o, s, Vs, lane);
    else pv_tiles<T, NKT>(o, s, Vs, lane);
#endif
    if constexpr (CLS) {
      if (qt == cls_tile) {   // the CLS row in fp32: the parts' output columns added up in lane c0 (both d halves: lanes c0 and c0 + 32)
        float* dst = cls_out + (int64_t)b * H * HD + h * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = o[dt][4 * rq + e];
              const float b1 = dpp_from_above<1>(a), b2 = NS == 3 ? dpp_from_above<2>(a) : 0.f;
              v[e] = cls_join<T>(a, b1, b2) * inv;
            }
            if (ql == c0) *(float4*)(dst + dt * 32 + 8 * rq + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
          }
      }
    }
    if constexpr (!HALF) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const uint32_t lo = pack2(o[dt][4 * rq] * inv, o[dt][4 * rq + 1] * inv, (T*)0);
          const uint32_t hi = pack2(o[dt][4 * rq + 2] * inv, o[dt][4 * rq + 3] * inv, (T*)0);
          *(u32x2*)(Ow + ql * 128 + (((dt * 4 + rq) ^ ((ql >> 1) & 7)) << 4) + g * 8) = mk2(lo, hi);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // DS ops of one wave complete in order; nothing else touches Ow
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = p * 8 + (lane >> 3), slot = lane & 7;
        const u32x4 v = *(const u32x4*)(Ow + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
        const int qq = qt * 32 + row;
        if (qq < L) store16_sc1(out + ((int64_t)b * L + qq) * H * HD + h * HD + slot * 8, v);
      }
      asm volatile("" ::: "memory");  // (compiler order only) these reads stay ahead of the next query tile's writes into Ow
    } else {  // the two 32-wide d halves one after the other, as 64-byte row pieces
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const uint32_t lo = pack2(o[dt][4 * rq] * inv, o[dt][4 * rq + 1] * inv, (T*)0);
          const uint32_t hi = pack2(o[dt][4 * rq + 2] * inv, o[dt][4 * rq + 3] * inv, (T*)0);
          *(u32x2*)(Ow + ql * 64 + ((rq ^ ((ql >> 2) & 3)) << 4) + g * 8) = mk2(lo, hi);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int row = p * 16 + (lane >> 2), slot = lane & 3;
          const u32x4 v = *(const u32x4*)(Ow + row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
          const int qq = qt * 32 + row;
          if (qq < L) store16_sc1(out + ((int64_t)b * L + qq) * H * HD + h * HD + dt * 32 + slot * 8, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  };

#ifdef ATTN_NO_KV_SPLIT
  bool v_pending = false;
#else
  bool v_pending = true;   // (wave-uniform) the V image has not been waited for yet: every wave takes that barrier exactly once
#endif
  while (has) {
    scores();
    if (v_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // V (and the prefetched Q fragments)
      __syncthreads();
      v_pending = false;
    }
    output();
    seq += 4;
    has = seq < ntile;
    qt = tile_at(min(seq, ntile - 1));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
  }
  if (v_pending) {   // a wave without a query tile (L <= 96)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// ================================================================================================
// temporal attention: one wave per (32 consecutive tokens, head); groups of Tn tokens
template <typename T>
__global__ __launch_bounds__(256) void attn_temporal_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, int64_t rows, int Tn,
                                                                int H, float scale, int64_t units, float* __restrict__ lse) {
  typedef AttnCfg<T> C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* Vs = smem + wave * 32 * C::RB;
  const int64_t ldq = 3 * (int64_t)H * HD;
  const int g = lane >> 5, ql = lane & 31;
  const int64_t iters = (units + (int64_t)gridDim.x * 4 - 1) / ((int64_t)gridDim.x * 4);
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t unit = (it * gridDim.x + blockIdx.x) * 4 + wave;
    const bool active = unit < units;
    const int64_t chunk = active ? unit / H : 0;
    const int h = active ? (int)(unit - chunk * H) : 0;
    const int64_t r0 = chunk * 32;
    const T* base = qkv + h * HD;
    // V tile -> wave-private LDS (coalesced: CPR lanes cover one 64-wide row)
    __syncthreads();  // previous iteration's column reads are done
#pragma unroll
    for (int i = 0; i < C::CPR / 2; ++i) {
      const int c = lane + i * 64, row = c / C::CPR, ch = c - row * C::CPR;
      u32x4 vv = mk4(0, 0, 0, 0);
      if (active && r0 + row < rows) vv = *(const u32x4*)(base + (r0 + row) * ldq + 2 * H * HD + ch * C::CN);
      *(u32x4*)(Vs + row * C::RB + (v_swz<T>(row, ch) << 4)) = vv;
    }
    const int64_t qrow = min(r0 + ql, rows - 1);
    u32x4 qf[C::KS], kf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const T* src = base + qrow * ldq + (2 * ks + g) * C::CN;
      qf[ks] = *(const u32x4*)src;
      kf[ks] = *(const u32x4*)(src + H * HD);
    }
    f32x16 s[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[0][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) mma_chunk<T>(s[0], kf[ks], qf[ks]);
    const int qgrp = ql / Tn;
    const float l_se = softmax_tiles<1>(s, scale, [&](int, int rq) {
      const int k0 = 8 * rq + 4 * g;  // block-diagonal mask: a query attends to the T frames of its own patch
      return make_float4((k0 + 0) / Tn == qgrp ? 0.f : -INFINITY, (k0 + 1) / Tn == qgrp ? 0.f : -INFINITY,
                         (k0 + 2) / Tn == qgrp ? 0.f : -INFINITY, (k0 + 3) / Tn == qgrp ? 0.f : -INFINITY);
    });
    if (lse && active && g == 0) lse[unit * 32 + ql] = l_se;  // (chunk*H + h)*32 + token
    __syncthreads();  // V tile visible
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    pv_tiles<T, 1>(o, s, Vs, lane);
    if (active && r0 + ql < rows) store_o<T>(out + (r0 + ql) * H * HD + h * HD, o, lane);
  }
}

// ================================================================================================
// 16-bit temporal attention, throughput form (round 2): one WAVE per (32 consecutive tokens, head) unit, everything wave-private
// like attn_temporal_bwd16.  The three 4 KiB tiles K, V, Q of the unit go global -> LDS by DMA in 128-byte row pieces (12 copies per
// unit, bank swizzle on the source side) instead of fragment-shaped 16-byte-per-row register loads (32 cache lines per instruction),
// log2-domain softmax with the 1/sum applied to the 32x64 output tile, and the output leaves through the dead K tile as 16-byte row
// stores.  No workgroup barrier: 4 independent waves per workgroup, units handed out grid-stride.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_temporal_fwd16_kernel(const T* __restrict__ qkv, T* __restrict__ out, int64_t rows, int Tn, int H,
                                                                    float scale, int64_t units, float* __restrict__ lse) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int WB = 3 * 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* tK = smem + wave * WB;
  char* tV = tK + 4096;
  char* tQ = tK + 8192;
  const uint32_t lds0 = lds_addr_of(tK);
  const char* zero = (const char*)g_attn_zero;
  const int64_t ldq = 3 * (int64_t)H * HD, ldo = (int64_t)H * HD;
  const int g = lane >> 5, ql = lane & 31;
  const float sl = scale * LOG2E;
  const int qgrp = ql / Tn;
  for (int64_t unit = (int64_t)blockIdx.x * 4 + wave; unit < units; unit += (int64_t)gridDim.x * 4) {
    const int64_t chunk = unit / H;
    const int h = (int)(unit - chunk * H);
    const int64_t r0 = chunk * 32;
    const int Le = (int)((rows - r0) < 32 ? (rows - r0) : 32);
    const T* qb = qkv + r0 * ldq + h * HD;
    // (the previous unit's LDS reads fed MFMAs / global stores that were issued before this point, so they have completed)
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) {
      const int row = piece * 8 + (lane >> 3), slot = lane & 7;
      const bool ok = row < Le;
      const T* src = qb + (int64_t)row * ldq;
      const int ck = (slot ^ ((row >> 1) & 7)) << 3, cv = (slot ^ (((row >> 1) & 1) << 2)) << 3;
      dma16(ok ? (const char*)(src + H * HD + ck) : zero, __builtin_amdgcn_readfirstlane(lds0 + piece * 1024));
      dma16(ok ? (const char*)(src + 2 * H * HD + cv) : zero, __builtin_amdgcn_readfirstlane(lds0 + 4096 + piece * 1024));
      dma16(ok ? (const char*)(src + ck) : zero, __builtin_amdgcn_readfirstlane(lds0 + 8192 + piece * 1024));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x16 s[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[0][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = ql * 128 + (((2 * ks + g) ^ ((ql >> 1) & 7)) << 4);
      mma_chunk<T>(s[0], *(const u32x4*)(tK + off), *(const u32x4*)(tQ + off));
    }
    // block-diagonal mask: a query attends to the Tn frames of its own patch (vit.py:146-157)
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool mine = ((r & 3) + 8 * (r >> 2) + 4 * g) / Tn == qgrp;
      s[0][r] = mine ? s[0][r] * sl : -INFINITY;
      m = fmaxf(m, s[0][r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = __builtin_amdgcn_exp2f(s[0][r] - m);  // exp2(-inf) == 0 off the diagonal block
      s[0][r] = pr;
      sum += pr;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lse && g == 0 && ql < Le) lse[unit * 32 + ql] = (m + __builtin_amdgcn_logf(sum)) * LN2;  // (chunk*H + h)*32 + token
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    pv_tiles<T, 1>(o, s, tV, lane);
    // O^T (lane = query, 4 consecutive d per register quad) -> row-major rows through the dead K tile
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint32_t lo = pack2(o[dt][4 * rq] * inv, o[dt][4 * rq + 1] * inv, (T*)0);
        const uint32_t hi = pack2(o[dt][4 * rq + 2] * inv, o[dt][4 * rq + 3] * inv, (T*)0);
        *(u32x2*)(tK + ql * 128 + (((dt * 4 + rq) ^ ((ql >> 1) & 7)) << 4) + g * 8) = mk2(lo, hi);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // DS ops of one wave complete in order; nothing else touches this tile
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = p * 8 + (lane >> 3), slot = lane & 7;
      const u32x4 v = *(const u32x4*)(tK + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      if (row < Le) store16_sc1(out + (r0 + row) * ldo + h * HD + slot * 8, v);
    }
  }
}

template <typename T, int NKT>
int launch_attn(const void* qkv, void* out, int batch, int L, int H, float scale, const float* key_bias, float* lse, float drop_p,
                uint32_t drop_seed, hipStream_t st) {
  const size_t lds = 2 * (size_t)NKT * 32 * AttnCfg<T>::RB + (size_t)NKT * 32 * sizeof(float);
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<T, NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL((attn_fwd_kernel<T, NKT>), dim3(batch * H), dim3(256), lds, st, (const T*)qkv, (T*)out, L, H, scale, key_bias, lse, drop_p, drop_seed);
  return check_launch("alpro_attn_fwd");
}

template <typename T, int NKT, bool HAS_BIAS>
int launch_attn16(const void* qkv, void* out, int batch, int L, int H, float scale, const float* key_bias, float* lse, float drop_p,
                  uint32_t drop_seed, const float* cls_q, int cls_group, float* cls_out, hipStream_t st) {
  const size_t lds = 2 * (size_t)NKT * 32 * 128 + 4 * (NKT == 8 ? 2048 : 4096) + (size_t)NKT * 32 * sizeof(float);
  static DeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)attn_fwd16_kernel<T, NKT, HAS_BIAS, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd16_kernel<T, NKT, HAS_BIAS, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd16_kernel<T, NKT, HAS_BIAS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
#define ALPRO_ATTN16_GO(CLS_, DROP_)                                                                                                                     \
  hipLaunchKernelGGL((attn_fwd16_kernel<T, NKT, HAS_BIAS, CLS_, DROP_>), dim3(batch * H), dim3(256), lds, st, (const T*)qkv, (T*)out, L, H, scale, key_bias, lse, drop_p, \
                     drop_seed, get_option(OPT_ATTN_ORDER), cls_q, cls_group, cls_out)
  // Three instantiations per (NKT, bias): plain, with the CLS parts, with the CLS parts and dropout.  A dropout launch WITHOUT a precise CLS
  // query runs the third one with cls_q == nullptr (the CLS code is skipped by a wave-uniform test): the <no CLS, dropout> form is the one
  // instantiation the register allocator does not get through without spilling (224-420 bytes of scratch at 7 / 8 key tiles, ROCm 7.2).
  if (drop_seed) ALPRO_ATTN16_GO(true, true);
#ifdef ATTN_CLS_FORCE_TPL  // (measurement builds: the instantiation with the CLS code on launches without a CLS query -- what does the code cost the regular rows?)
  else if (true) ALPRO_ATTN16_GO(true, false);
#else
  else if (cls_q) ALPRO_ATTN16_GO(true, false);
#endif
  else ALPRO_ATTN16_GO(false, false);
#undef ALPRO_ATTN16_GO
  return check_launch("alpro_attn_fwd");
}

template <typename T>
int dispatch_attn(const void* qkv, void* out, int batch, int L, int H, float scale, const float* key_bias, float* lse, float dp, uint32_t ds,
                  const float* cls_q, int cls_group, float* cls_out, hipStream_t st) {
  const int nkt = (L + 31) / 32;
  if constexpr (sizeof(T) == 2) {
#define ALPRO_ATTN16(N)                                                                                       \
  return key_bias ? launch_attn16<T, N, true>(qkv, out, batch, L, H, scale, key_bias, lse, dp, ds, cls_q, cls_group, cls_out, st)         \
                  : launch_attn16<T, N, false>(qkv, out, batch, L, H, scale, key_bias, lse, dp, ds, cls_q, cls_group, cls_out, st)
    if (nkt <= 2) ALPRO_ATTN16(2);
    if (nkt <= 4) ALPRO_ATTN16(4);
    if (nkt <= 7) ALPRO_ATTN16(7);
    ALPRO_ATTN16(8);
#undef ALPRO_ATTN16
  }
  if (nkt <= 2) return launch_attn<T, 2>(qkv, out, batch, L, H, scale, key_bias, lse, dp, ds, st);
  if (nkt <= 4) return launch_attn<T, 4>(qkv, out, batch, L, H, scale, key_bias, lse, dp, ds, st);
  if (nkt <= 7) return launch_attn<T, 7>(qkv, out, batch, L, H, scale, key_bias, lse, dp, ds, st);
  return launch_attn<T, 8>(qkv, out, batch, L, H, scale, key_bias, lse, dp, ds, st);
}

}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" int alpro_attn_fwd(const void* qkv, void* out, int dtype, int batch, int L, int H, float scale,
                              const float* key_bias, float* lse, float drop_p, uint32_t drop_seed, const float* cls_q, int cls_group, float* cls_out,
                              void* stream) {
  ALPRO_CHECK(!cls_q || (cls_out && cls_group > 0 && batch % cls_group == 0 && dtype != ALPRO_F32 && ((uintptr_t)cls_q % 16) == 0 && ((uintptr_t)cls_out % 16) == 0),
              "alpro_attn_fwd: the precise CLS query needs cls_out, a group size dividing the batch, 16-byte aligned fp32 side tensors and a 16-bit operand dtype");
  ALPRO_CHECK(qkv && out && batch > 0 && H > 0, "alpro_attn_fwd: bad args");
  ALPRO_CHECK(L > 0 && L <= 256, "alpro_attn_fwd: L=%d unsupported (1..256; the path needs 40, 197, 237)", L);
  ALPRO_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "alpro_attn_fwd: pointers must be 16-byte aligned");
  ALPRO_CHECK(!drop_seed || (drop_p > 0.f && drop_p < 1.f), "alpro_attn_fwd: dropout needs 0 < p < 1");
  ALPRO_DISPATCH_DTYPE(dtype, T, return dispatch_attn<T>(qkv, out, batch, L, H, scale, key_bias, lse, drop_p, drop_seed, cls_q, cls_group, cls_out, (hipStream_t)stream));
  return ALPRO_OK;
}

extern "C" int alpro_attn_temporal_fwd(const void* qkv, void* out, int dtype, int64_t rows, int T, int H, float scale, float* lse, void* stream) {
  ALPRO_CHECK(qkv && out && rows > 0 && H > 0, "alpro_attn_temporal_fwd: bad args");
  ALPRO_CHECK(T > 0 && 32 % T == 0, "alpro_attn_temporal_fwd: num_frm=%d must divide 32", T);
  ALPRO_CHECK(rows % T == 0, "alpro_attn_temporal_fwd: rows=%lld not a multiple of T=%d", (long long)rows, T);
  ALPRO_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "alpro_attn_temporal_fwd: pointers must be 16-byte aligned");
  const int64_t units = ((rows + 31) / 32) * H;
  int64_t grid = (units + 3) / 4;
  if (grid > 256 * 8) grid = 256 * 8;
  if (dtype != ALPRO_F32) {  // 16-bit storage: DMA-staged wave-private tiles
    const size_t lds16 = 4 * 3 * 4096;
    static DeviceOnce attr_once;
    attr_once.run([&] {
      (void)hipFuncSetAttribute((const void*)attn_temporal_fwd16_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
      (void)hipFuncSetAttribute((const void*)attn_temporal_fwd16_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
    });
    if (grid > 512) grid = 512;  // 2 workgroups per CU, units handed out grid-stride
    if (dtype == ALPRO_BF16)
      hipLaunchKernelGGL(attn_temporal_fwd16_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), lds16, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, rows, T, H, scale, units, lse);
    else
      hipLaunchKernelGGL(attn_temporal_fwd16_kernel<f16_t>, dim3((unsigned)grid), dim3(256), lds16, (hipStream_t)stream, (const f16_t*)qkv, (f16_t*)out, rows, T, H, scale, units, lse);
    return check_launch("alpro_attn_temporal_fwd");
  }
  const int esz = dtype == ALPRO_F32 ? 4 : 2;
  const size_t lds = 4 * 32 * 64 * (size_t)esz;
  ALPRO_DISPATCH_DTYPE(dtype, T_, hipLaunchKernelGGL(attn_temporal_fwd_kernel<T_>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, (const T_*)qkv, (T_*)out, rows, T, H, scale, units, lse));
  return check_launch("alpro_attn_temporal_fwd");
}
