// Shared device/host helpers for the ALPRO gfx950 kernels.  CDNA4 only: wave64, MFMA 32x32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alpro_hip.h"

namespace alpro {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// native vector types for register-resident 16-/8-byte chunks (HIP's uint4 struct defeats SROA in
// staged pipelines and lands in scratch)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 mk4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return u32x4{a, b, c, d}; }
__device__ __forceinline__ u32x2 mk2(uint32_t a, uint32_t b) { return u32x2{a, b}; }
// by-value bit casts: __builtin_bit_cast applied directly to an ext-vector element lvalue (v.y)
// reads element 0 on this toolchain, so always go through a scalar rvalue.
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct bf16_t { uint16_t v; };  // storage-only 16-bit types (arithmetic is always fp32)
struct f16_t { uint16_t v; };

// ---- scalar conversions (round-to-nearest-even, NaN preserved) -------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return __builtin_bit_cast(float, (uint32_t)x.v << 16); }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)__builtin_bit_cast(_Float16, x.v); }

// hardware conversion (v_cvt_pk_bf16_f32 on gfx950: round-to-nearest-even, NaN preserved), branch-free
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return bf16_t{f32_to_bf16_bits(x)}; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) {
  return f16_t{__builtin_bit_cast(uint16_t, (_Float16)x)};
}

// Values as the MFMA will see them: round-trip through the storage type.
template <typename T> __device__ __forceinline__ float quantize(float x) { return to_f32(from_f32<T>(x)); }

// ---- 16-byte "k-chunk": the unit every tile/fragment path moves -------------------------------
// bf16/f16: 8 elements = one K=16 MFMA operand per lane; f32: 4 elements = four K=2 MFMA operands.
template <typename T> struct Chunk { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void unpack_chunk(const u32x4& c, float (&o)[Chunk<T>::N]);
template <> __device__ __forceinline__ void unpack_chunk<float>(const u32x4& c, float (&o)[4]) {
  o[0] = u2f(c.x); o[1] = u2f(c.y);
  o[2] = u2f(c.z); o[3] = u2f(c.w);
}
template <> __device__ __forceinline__ void unpack_chunk<bf16_t>(const u32x4& c, float (&o)[8]) {
  const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = u2f(w[i] << 16);
    o[2 * i + 1] = u2f(w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void unpack_chunk<f16_t>(const u32x4& c, float (&o)[8]) {
  const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] & 0xffffu));
    o[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] >> 16));
  }
}

template <typename T> __device__ __forceinline__ void unpack_pair(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack_pair<bf16_t>(uint32_t w, float& lo, float& hi) {
  lo = u2f(w << 16);
  hi = u2f(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack_pair<f16_t>(uint32_t w, float& lo, float& hi) {
  lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
  hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float lo, float hi, bf16_t*) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi, f16_t*) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, f16x2_t));
}
template <typename T> __device__ __forceinline__ u32x4 pack_chunk(const float* v);
template <> __device__ __forceinline__ u32x4 pack_chunk<float>(const float* v) {
  return mk4(f2u(v[0]), f2u(v[1]), f2u(v[2]), f2u(v[3]));
}
template <> __device__ __forceinline__ u32x4 pack_chunk<bf16_t>(const float* v) {
  return mk4(pack2(v[0], v[1], (bf16_t*)0), pack2(v[2], v[3], (bf16_t*)0), pack2(v[4], v[5], (bf16_t*)0),
                    pack2(v[6], v[7], (bf16_t*)0));
}
template <> __device__ __forceinline__ u32x4 pack_chunk<f16_t>(const float* v) {
  return mk4(pack2(v[0], v[1], (f16_t*)0), pack2(v[2], v[3], (f16_t*)0), pack2(v[4], v[5], (f16_t*)0),
                    pack2(v[6], v[7], (f16_t*)0));
}

// ---- MFMA on one pair of 16-byte chunks: acc(32x32) += A(32 x kc) * B(kc x 32) ----------------
// Lane l supplies row (l & 31) of its operand and the k-slice selected by (l >> 5); both operands
// use the same slice order, so any consistent k assignment is correct.
template <typename T> __device__ __forceinline__ void mma_chunk(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma_chunk<bf16_t>(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<f16_t>(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(a.x), u2f(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(a.y), u2f(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(a.z), u2f(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(a.w), u2f(b.w), acc, 0, 0, 0);
}
// Row of a 32x32 accumulator register r held by lane l (column is l & 31): CDNA C/D layout.
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- wave64 reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// 0.5 * erfc(z) for z >= 0 as ONE exponential: erfc(z) ~= 2^q(z), q a degree-5 polynomial without constant term fitted on
// [0, 5.5] (max abs error 6.8e-7; beyond 5.5 the polynomial keeps falling and 2^q underflows to 0 like erfc does).
// No reciprocal and no second transcendental: 5 FMA-class ops + v_exp_f32, vs rcp + exp + 7 for an Abramowitz-Stegun rational form.
__device__ __forceinline__ float half_erfc_pos(float z) {
  float q = fmaf(z, -0.00296695f, 0.02966973f);
  q = fmaf(z, q, -0.14875628f);
  q = fmaf(z, q, -0.91847146f);
  q = fmaf(z, q, -1.62789348f);
  return __builtin_amdgcn_exp2f(fmaf(z, q, -1.0f));
}
// exact-mode (f32 storage) keeps libm erff; 16-bit storage: gelu(x) = x * Phi(x), Phi(x) = 0.5 erfc(-x / sqrt 2),
// |error| <= 1.2e-6 absolute (three orders below bf16 resolution at |x| ~ 1)
template <typename T> __device__ __forceinline__ float gelu_fast(float x) {
  if constexpr (sizeof(T) == 4) return gelu_erf(x);
  else {
    const float h = half_erfc_pos(fabsf(x) * 0.70710678118654752440f);  // Phi(-|x|)
    return x * (0.5f + copysignf(0.5f - h, x));
  }
}

// gelu'(x) = Phi(x) + x phi(x)
template <typename T> __device__ __forceinline__ float gelu_grad(float x) {
  if constexpr (sizeof(T) == 4) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
  } else {
    const float h = half_erfc_pos(fabsf(x) * 0.70710678118654752440f);
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2)
    return (0.5f + copysignf(0.5f - h, x)) + x * pdf;
  }
}

// gelu(x) and gelu'(x) for TWO elements with ONE exponential each (round 5).  Phi(-|x|) = 0.5 erfc(|x| / sqrt 2) = exp(-x^2 / 2) * g(|x|) with
// g(t) = 0.5 erfcx(t / sqrt 2), and exp(-x^2 / 2) is what phi(x) needs anyway: g is smooth and slowly varying, a degree-9 polynomial in |x|
// (weighted minimax fit, the weight being the exponential in front of it: |Phi error| <= 3.2e-7 in fp32 arithmetic on [0, 13], checked against
// scipy's erfc in the fit script's output recorded in DESIGN.md) replaces the second v_exp_f32 of the round-3 form (erfc ~= 2^q(z) plus a
// separate exp for phi).  Written on 2-vectors so that the nine Horner steps are v_pk_fma_f32 (two elements per issue slot): per pair of
// elements 9 + ~10 VALU slots and 2 transcendentals, against 25 + 4.  The GELU-saving fc1 epilogue spent about 40 % of a tile's time here.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_and_grad2(f32x2v x, f32x2v& y, f32x2v& dy) {
  const f32x2v t = {fabsf(x.x), fabsf(x.y)};
  f32x2v p = t * (f32x2v){-8.062383613e-06f, -8.062383613e-06f} + (f32x2v){1.519315725e-04f, 1.519315725e-04f};
  p = p * t + (f32x2v){-1.271098853e-03f, -1.271098853e-03f};
  p = p * t + (f32x2v){6.382599637e-03f, 6.382599637e-03f};
  p = p * t + (f32x2v){-2.223049180e-02f, -2.223049180e-02f};
  p = p * t + (f32x2v){5.949637600e-02f, 5.949637600e-02f};
  p = p * t + (f32x2v){-1.317726293e-01f, -1.317726293e-01f};
  p = p * t + (f32x2v){2.497522130e-01f, 2.497522130e-01f};
  p = p * t + (f32x2v){-3.989226083e-01f, -3.989226083e-01f};
  p = p * t + (f32x2v){4.999997430e-01f, 4.999997430e-01f};
  const f32x2v a = x * x * (f32x2v){-0.72134752044448170368f, -0.72134752044448170368f};   // -x^2 / 2 in the log2 domain
  const f32x2v e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};           // exp(-x^2 / 2)
  const f32x2v h = e * p;                                                                // Phi(-|x|)
  const f32x2v cdf = {0.5f + copysignf(0.5f - h.x, x.x), 0.5f + copysignf(0.5f - h.y, x.y)};
  y = x * cdf;
  dy = cdf + x * (e * (f32x2v){0.39894228040143267794f, 0.39894228040143267794f});
}

// gelu(x) alone on two elements: gelu_fast's arithmetic (erfc ~= 2^q(z), one exponential per element) with the polynomial on 2-vectors
__device__ __forceinline__ f32x2v gelu_fast2(f32x2v x) {
  const f32x2v z = (f32x2v){fabsf(x.x), fabsf(x.y)} * (f32x2v){0.70710678118654752440f, 0.70710678118654752440f};
  f32x2v q = z * (f32x2v){-0.00296695f, -0.00296695f} + (f32x2v){0.02966973f, 0.02966973f};
  q = z * q + (f32x2v){-0.14875628f, -0.14875628f};
  q = z * q + (f32x2v){-0.91847146f, -0.91847146f};
  q = z * q + (f32x2v){-1.62789348f, -1.62789348f};
  q = z * q + (f32x2v){-1.0f, -1.0f};
  const f32x2v h = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  const f32x2v cdf = {0.5f + copysignf(0.5f - h.x, x.x), 0.5f + copysignf(0.5f - h.y, x.y)};
  return x * cdf;
}

// gelu(x) and gelu'(x) together (the forward of a GELU Linear that keeps gelu' for its backward): Phi(x) is shared
template <typename T> __device__ __forceinline__ void gelu_and_grad(float x, float& y, float& dy) {
  if constexpr (sizeof(T) == 4) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    y = x * cdf;
    dy = cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
  } else {
    const float h = half_erfc_pos(fabsf(x) * 0.70710678118654752440f);
    const float cdf = 0.5f + copysignf(0.5f - h, x);
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
    y = x * cdf;
    dy = cdf + x * pdf;
  }
}

// ---- counter-based dropout mask: a pure function of (seed, element index), so the backward regenerates it ------
__device__ __forceinline__ uint32_t mix32(uint32_t h) {  // "lowbias32" finalizer
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h;
}
// keep probability 1 - p: keep iff the top 24 hash bits >= thresh24 = round(p * 2^24)
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint64_t idx, uint32_t thresh24) {
  const uint32_t h = mix32(seed ^ ((uint32_t)idx * 0x9E3779B1u) ^ ((uint32_t)(idx >> 32) * 0x85EBCA77u));
  return (h >> 8) >= thresh24;
}
__host__ __device__ inline uint32_t drop_thresh24(float p) { return (uint32_t)(p * 16777216.0f + 0.5f); }

// ---- global -> LDS DMA issued from inline asm ---------------------------------------------------------------
// hipcc makes every LDS read wait for ALL outstanding global_load_lds it knows about (it cannot prove that the
// buffer being filled is not the one being read), which serialises copy and compute.  Hidden from the compiler,
// the copies are tracked by hand with s_waitcnt vmcnt(N).  lds_addr: wave-uniform LDS byte address of the 1 KiB
// piece; lane l's 16 bytes land at lds_addr + 16 l.
__device__ __forceinline__ void dma16(const void* src, uint32_t lds_addr) {
  // reserved-register site (the product is built with -Werror=inline-asm; this one is deliberate): global_load_lds takes its LDS address from m0; listing it as clobbered is what keeps the compiler from assuming a value of its own survives the statement (it writes m0 itself before each of its own uses: LDS-DMA builtins, s_movrel); the K-loop ISA tests of tests/test_host_cpu.py read the built object
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
#pragma clang diagnostic pop
}
// 16-byte write-through store (sc1): the line leaves the XCD's L2 and lands memory-side.  Used for the attention outputs (read next by
// the projection GEMM / the weight gradient): same speed as a non-temporal store for the attention kernel, ~0.3 % of the step for its
// consumers (profiles/r2_gemm_epilogue_experiments.txt, experiment 10d); on the GEMM's own 16-bit outputs it LOSES 1.8 ms.
// (the trailing s_nop 1: a store of more than 8 bytes reads its data registers a moment after it issues, and the compiler -- which sees an
// opaque string, not a store -- may let the very next VALU instruction overwrite them; round 6: the q | k | v rows of the fused temporal kernel
// came out wrong in some builds and bitwise right in others until the wait states were inside the string)
__device__ __forceinline__ void store16_sc1(void* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// A SCALAR load of read-only data at a wave-uniform index (s_load_dword: counted on lgkmcnt, not on vmcnt -- it cannot put a vmcnt(0) join
// between an epilogue's output stores).  The constant address space is what makes the compiler pick the scalar path.
__device__ __forceinline__ float sload_f32(const float* p, uint32_t idx) {
  return *((const __attribute__((address_space(4))) float*)p + __builtin_amdgcn_readfirstlane(idx));
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

// ---- host side -----------------------------------------------------------------------------
void set_error(const char* fmt, ...);

// Tuning knobs (measurement aids, not part of the arithmetic): initialised ONCE from the environment when the library is
// loaded (ALPRO_GEMM_TILE / ALPRO_GEMM_GRID / ALPRO_GEMM_TUNE / ALPRO_TN_SPLITS), changed at run time only through
// alpro_hip_set_option -- no getenv() on the launch path.  0 = "not set" for every knob except GEMM_TUNE.
// Unit order of the per-(sequence, head) 16-bit attention kernels.  Workgroups go to the 8 XCDs round-robin (blockIdx % 8), each XCD with
// its own L2.  In the identity order the 12 heads of a sequence -- adjacent 128-byte pieces of the same 4608-byte rows -- are spread over
// 8 XCDs; order 1 gives XCD x the whole sequences b = x (mod 8), their heads in dispatch order (a bijection when the batch is a multiple of 8;
// otherwise the identity order is kept).
__device__ __forceinline__ void attn_unit(int blk, int nblk, int H, int order, int& b, int& h) {
  const int batch = nblk / H;
  if (order == 1 && (batch & 7) == 0) {
    const int x = blk & 7, i = blk >> 3, j = i / H;
    h = i - j * H;
    b = j * 8 + x;
  } else {
    b = blk / H;
    h = blk - b * H;
  }
}
enum { OPT_GEMM_TILE = 0, OPT_GEMM_GRID = 1, OPT_GEMM_TUNE = 2, OPT_TN_SPLITS = 3, OPT_TN_KIND = 4, OPT_GEMM_TAIL = 5, OPT_ATTN_BWD = 6, OPT_ATTN_ORDER = 7, OPT_GEMM_KIND = 8, OPT_CU_BUDGET = 9, OPT_LN_GRID = 10, OPT_GEMM_SCHED = 11, OPT_GEMM_EPI = 12, OPT_COUNT = 13 };
int get_option(int which);
// Per-stream override of an option (round 5: alpro_hip_set_stream_option).  cu_budget is the one knob that describes the SITUATION a launch
// runs in (a collective's kernels holding CUs next to this stream's work) rather than the library: the optimizer sets it on the stream its
// backward runs on, other streams / other models in the same process keep the process-wide value.  -1 = no override for this stream.
int get_stream_option(int which, hipStream_t st);

// ---- dynamic tile scheduler of the persistent GEMM grids (round 5) ------------------------------------------------------------------
// One block of device memory per launch: [0..7] per-XCD ticket counters, [SCHED_CLAIM0 + w] the claim word of workgroup w (its two static
// tiles).  Every (device, stream) owns a PAIR of blocks used alternately: launch n works on block n & 1 and its first workgroup zeroes the
// other one -- the block of launch n - 1, which is complete because launches of one stream are ordered.  No workgroup ever waits for a
// "last one out" count, and no block is shared between streams.
constexpr int SCHED_CLAIM0 = 8, SCHED_MAX_WG = 256, SCHED_BLOCK_U32 = SCHED_CLAIM0 + SCHED_MAX_WG;
struct TileSched {
  uint32_t* blk;       // nullptr = static walk (option gemm_sched 0, a stream that is being captured, or no block pair left)
  uint32_t* prev;      // the block of this stream's previous launch, zeroed by workgroup 0 (never nullptr when blk is set)
  uint32_t magic_ntn;  // ceil(2^32 / ntn): tile / ntn == umulhi(tile, magic_ntn) (ntn > 1)
  uint32_t epi;        // option gemm_epi: 1 = packed 16-bit epilogue where it applies (default), 0 = the staged fp32 epilogue everywhere
};
// Picks the stream's next block and keeps the stream's slot locked until the destructor runs, i.e. until the caller has enqueued the launch:
// two host threads launching on one stream cannot interleave "pick" and "enqueue" in opposite orders.
struct SchedLaunch {
  uint32_t* cur = nullptr;
  uint32_t* prev = nullptr;
  void* slot = nullptr;
  SchedLaunch(hipStream_t st, bool enabled);
  void commit(bool launched);   // call once, right after the launch: advances the pair's parity only if the launch was accepted
  ~SchedLaunch();
};
int sched_set_workspace(hipStream_t st, void* ptr, size_t bytes);
inline uint32_t magic_u32(uint32_t d) { return (uint32_t)(((1ull << 32) + d - 1) / d); }
// Compute units the persistent GEMM grids and the weight-gradient range plan may count on (round 4): 256, or less while a collective's kernels
// hold CUs (alpro_amd.dist sets "cu_budget" while the overlapped gradient exchange is in flight: a persistent one-workgroup-per-CU launch that
// finds a CU taken needs a second round for the displaced workgroup).  Rounded down to a multiple of 8 (XCD-contiguous slot maps).
inline int cu_budget(hipStream_t st) {
  int v = get_stream_option(OPT_CU_BUDGET, st);
  if (v < 0) v = get_option(OPT_CU_BUDGET);
  return (v >= 64 && v < 256) ? (v / 8 * 8) : 256;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) must be applied once per (kernel, device): a per-instantiation bit mask
// over device ordinals, safe under concurrent first launches (setting the attribute twice is harmless).
struct DeviceOnce {
  unsigned long long mask = 0;
  template <typename F> void run(F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & bit)) {
      f();
      __atomic_fetch_or(&mask, bit, __ATOMIC_RELEASE);
    }
  }
};
#define ALPRO_CHECK(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      alpro::set_error(__VA_ARGS__);      \
      return ALPRO_ERR_INVALID;           \
    }                                     \
  } while (0)
int check_launch(const char* what);

// dispatch a templated launcher on the storage dtype
#define ALPRO_DISPATCH_DTYPE(dt, T, ...)                                         \
  switch (dt) {                                                                  \
    case ALPRO_F32: { typedef float T; __VA_ARGS__; break; }                     \
    case ALPRO_BF16: { typedef alpro::bf16_t T; __VA_ARGS__; break; }            \
    case ALPRO_F16: { typedef alpro::f16_t T; __VA_ARGS__; break; }              \
    default: alpro::set_error("bad dtype %d", (int)(dt)); return ALPRO_ERR_INVALID; \
  }

}  // namespace alpro
