// Error plumbing + small memory-bound kernels (casts, patchify, CLS mean, BERT embeddings,
// final-norm temporal pooling).  All are HBM-bound: one wave per 768-wide row, 16-byte accesses.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace alpro {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
const char* const kOptNames[OPT_COUNT] = {"gemm_tile", "gemm_grid", "gemm_tune", "tn_splits", "tn_kind", "gemm_tail", "attn_bwd", "attn_order", "gemm_kind", "cu_budget", "ln_grid", "gemm_sched", "gemm_epi"};
const char* const kOptEnv[OPT_COUNT] = {"ALPRO_GEMM_TILE", "ALPRO_GEMM_GRID", "ALPRO_GEMM_TUNE", "ALPRO_TN_SPLITS", "ALPRO_TN_KIND", "ALPRO_GEMM_TAIL", "ALPRO_ATTN_BWD", "ALPRO_ATTN_ORDER", "ALPRO_GEMM_KIND", "ALPRO_CU_BUDGET", "ALPRO_LN_GRID", "ALPRO_GEMM_SCHED", "ALPRO_GEMM_EPI"};
int g_opts[OPT_COUNT];
// Values that change RESULTS (gemm_tune 3 / 4 / 10 / 11 / 12: epilogue / synchronisation ablations; tn_kind 1: weight gradient without
// its epilogue) exist only in the measurement build (-DALPRO_ABLATIONS, `python -m alpro_amd.build --ablations`, used by tools/); the
// product library refuses them, from alpro_hip_set_option and from the environment alike.
bool option_allowed(int which, int value) {
#ifdef ALPRO_ABLATIONS
  (void)which; (void)value;
  return true;
#else
  if (which == OPT_GEMM_TUNE) return value >= 0 && value <= 2;
  if (which == OPT_TN_KIND) return value == 0 || value == 2;   // 1: no epilogue (timing only, measurement build); 2: two-group schedule
  if (which == OPT_ATTN_BWD) return value >= 0 && value <= 2;   // 3 / 4: the persistent key-owned backward, compiled into the measurement build only (round 4)
  return value >= 0;
#endif
}
struct OptInit {
  OptInit() {
    for (int i = 0; i < OPT_COUNT; ++i) {
      const char* e = getenv(kOptEnv[i]);
      const int dflt = (i == OPT_GEMM_TUNE || i == OPT_GEMM_TAIL || i == OPT_ATTN_BWD || i == OPT_GEMM_KIND || i == OPT_GEMM_SCHED || i == OPT_GEMM_EPI) ? 1
                       : i == OPT_TN_KIND ? 2   // (round 5: the two-group schedule of the weight-gradient kernel, -1.2 % on its 29.7 ms in the step: profiles/r5_tn_kind_ab.txt)
                                          : 0;
      g_opts[i] = e ? atoi(e) : dflt;
      if (!option_allowed(i, g_opts[i])) {
        fprintf(stderr, "libalpro_hip: %s=%d is a result-corrupting ablation and is not part of this build (ignored)\n", kOptEnv[i], g_opts[i]);
        g_opts[i] = dflt;
      }
    }
  }
} g_opt_init;
}  // namespace

int get_option(int which) { return __atomic_load_n(&g_opts[which], __ATOMIC_RELAXED); }

// ---- per-stream option overrides (round 5) ---------------------------------------------------------------------------------------
// A handful of (stream, option) -> value entries behind a spin lock: set by the owner of a stream (alpro_amd.optim while its gradient
// exchange is in flight), read once per launch that consults the option.  An empty table (the normal case) costs one relaxed load.
namespace {
struct StreamOpt { hipStream_t st; int which, value; };
constexpr int kMaxStreamOpts = 32;
StreamOpt g_sopts[kMaxStreamOpts];
int g_nsopts = 0;
int g_sopt_lock = 0;
struct SpinGuard {
  SpinGuard() { while (__atomic_exchange_n(&g_sopt_lock, 1, __ATOMIC_ACQUIRE)) {} }
  ~SpinGuard() { __atomic_store_n(&g_sopt_lock, 0, __ATOMIC_RELEASE); }
};
}  // namespace

int get_stream_option(int which, hipStream_t st) {
  if (__atomic_load_n(&g_nsopts, __ATOMIC_RELAXED) == 0) return -1;
  SpinGuard g;
  for (int i = 0; i < g_nsopts; ++i)
    if (g_sopts[i].st == st && g_sopts[i].which == which) return g_sopts[i].value;
  return -1;
}

// ---- tile-scheduler blocks (common.hpp) ---------------------------------------------------------------------------------------------
// Who owns the memory.  Each (device, stream) that launches the persistent 8-phase GEMM needs one PAIR of counter blocks (2 x 1056 bytes).
//   * A caller that wants the library to allocate nothing registers its own: alpro_hip_set_sched_workspace(stream, ptr, bytes).
//   * Otherwise the first such launch on a device makes ONE hipMalloc of kSchedSlots pairs (135 KB) for that device -- the only allocation
//     this library ever makes -- and the stream's pair is cleared by a hipMemsetAsync ON THAT STREAM (ordered before the launch; no device-wide
//     synchronisation, round 6).
// A slot belongs to its (device, stream) until alpro_hip_release_stream(stream): a destroyed stream's handle value may be handed out again
// by the runtime, so owners of short-lived streams release them (the re-created stream then starts from a cleared pair instead of
// inheriting one).  With all kSchedSlots slots taken a further stream runs the static walk (same results, no CU-theft tolerance).
namespace {
constexpr int kSchedSlots = 64;
struct SchedSlot {
  int dev;
  hipStream_t st;
  uint32_t* pair;   // 2 x SCHED_BLOCK_U32 dwords
  unsigned n;       // launches committed on this pair: launch n works on block n & 1 and zeroes the other
  int lock;
  bool used, caller_owned;
};
SchedSlot g_sched[kSchedSlots];
int g_sched_table_lock = 0;
uint32_t* g_sched_pool[64];   // per device: kSchedSlots pairs (slot i's pair is pool + i * pair size)
constexpr size_t kPairBytes = (size_t)2 * SCHED_BLOCK_U32 * sizeof(uint32_t);
struct TableGuard {
  TableGuard() { while (__atomic_exchange_n(&g_sched_table_lock, 1, __ATOMIC_ACQUIRE)) {} }
  ~TableGuard() { __atomic_store_n(&g_sched_table_lock, 0, __ATOMIC_RELEASE); }
};
SchedSlot* find_slot(int dev, hipStream_t st) {
  for (int i = 0; i < kSchedSlots; ++i)
    if (g_sched[i].used && g_sched[i].dev == dev && g_sched[i].st == st) return &g_sched[i];
  return nullptr;
}
// a free slot for (dev, st) on `pair` (nullptr: the device pool's), cleared on the stream; nullptr when the table is full / the pool cannot be made
SchedSlot* make_slot(int dev, hipStream_t st, uint32_t* pair) {
  int at = -1;
  for (int i = 0; i < kSchedSlots && at < 0; ++i)
    if (!g_sched[i].used) at = i;
  if (at < 0) return nullptr;
  const bool own = pair != nullptr;
  if (!own) {
    if (!g_sched_pool[dev]) {
      void* pmem = nullptr;
      if (hipMalloc(&pmem, kSchedSlots * kPairBytes) == hipSuccess) g_sched_pool[dev] = (uint32_t*)pmem;
      else (void)hipGetLastError();
    }
    if (!g_sched_pool[dev]) return nullptr;
    pair = g_sched_pool[dev] + (size_t)at * 2 * SCHED_BLOCK_U32;
  }
  if (hipMemsetAsync(pair, 0, kPairBytes, st) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  SchedSlot* s = &g_sched[at];
  s->dev = dev;
  s->st = st;
  s->pair = pair;
  s->n = 0;
  s->lock = 0;
  s->caller_owned = own;
  s->used = true;
  return s;
}
}  // namespace

SchedLaunch::SchedLaunch(hipStream_t st, bool enabled) {
  if (!enabled) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {   // a captured launch is replayed with the SAME block: static walk
    (void)hipGetLastError();
    return;
  }
  SchedSlot* s = nullptr;
  {
    TableGuard g;
    s = find_slot(dev, st);
    if (!s) s = make_slot(dev, st, nullptr);
  }
  if (!s) return;
  while (__atomic_exchange_n(&s->lock, 1, __ATOMIC_ACQUIRE)) {}
  slot = s;
  cur = s->pair + (size_t)(s->n & 1u) * SCHED_BLOCK_U32;
  prev = s->pair + (size_t)((s->n + 1u) & 1u) * SCHED_BLOCK_U32;
}

// The pair's parity moves on only when the launch was accepted: a launch that failed at enqueue never ran, so `cur` is still
// clear and `prev` still holds the previous launch's counts -- exactly what the next attempt expects.
void SchedLaunch::commit(bool launched) {
  if (slot && launched) ++((SchedSlot*)slot)->n;
}

SchedLaunch::~SchedLaunch() {
  if (slot) __atomic_store_n(&((SchedSlot*)slot)->lock, 0, __ATOMIC_RELEASE);
}

int sched_set_workspace(hipStream_t st, void* ptr, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("alpro_hip_set_sched_workspace: no current device"); return ALPRO_ERR_INVALID; }
  TableGuard g;
  SchedSlot* s = find_slot(dev, st);
  if (s) {   // (the stream's launches so far are ordered before the clear make_slot() issues on it)
    while (__atomic_exchange_n(&s->lock, 1, __ATOMIC_ACQUIRE)) {}
    s->used = false;
    __atomic_store_n(&s->lock, 0, __ATOMIC_RELEASE);
  }
  if (!ptr) return ALPRO_OK;
  if (bytes < kPairBytes || ((uintptr_t)ptr % 16) != 0) { set_error("alpro_hip_set_sched_workspace: needs %zu bytes, 16-byte aligned (got %zu)", kPairBytes, bytes); return ALPRO_ERR_INVALID; }
  if (!make_slot(dev, st, (uint32_t*)ptr)) { set_error("alpro_hip_set_sched_workspace: all %d (device, stream) slots are taken: release a stream first", kSchedSlots); return ALPRO_ERR_INVALID; }
  return ALPRO_OK;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return ALPRO_ERR_LAUNCH;
  }
  return ALPRO_OK;
}

namespace {

// ---- cast ------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *(const float4*)(src + i), b = *(const float4*)(src + i + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      if constexpr (sizeof(T) == 4) {
        *(float4*)(dst + i) = a;
        *(float4*)(dst + i + 4) = b;
      } else {
        *(u32x4*)(dst + i) = pack_chunk<T>(v);
      }
    } else {
      for (int64_t k = i; k < n; ++k) dst[k] = from_f32<T>(src[k]);
    }
  }
}

// ---- patchify: im2col rows of the 16x16/stride-16 conv -----------------------------------------
// one thread per 16-byte output chunk... (8 or 4 consecutive j of one (c, i) patch row = contiguous pixels)
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int BT, int C, int H, int W) {
  constexpr int E = Chunk<T>::N;  // elements per 16-byte chunk
  const int gw = W / 16, gh = H / 16, N = gh * gw, K = C * 256;
  const int chunks_per_row = K / E;
  const int64_t total = (int64_t)BT * N * chunks_per_row;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = id / chunks_per_row;
    const int kc = (int)(id - row * chunks_per_row) * E;  // k = c*256 + i*16 + j
    const int bt = (int)(row / N), n = (int)(row - (int64_t)bt * N);
    const int ph = n / gw, pw = n - ph * gw;
    const int c = kc >> 8, i = (kc >> 4) & 15, j = kc & 15;
    const float* src = img + (((int64_t)bt * C + c) * H + (ph * 16 + i)) * W + pw * 16 + j;
    float v[E];
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const float4 f = *(const float4*)(src + e);
      v[e] = f.x; v[e + 1] = f.y; v[e + 2] = f.z; v[e + 3] = f.w;
    }
    *(u32x4*)(out + row * K + kc) = pack_chunk<T>(v);
  }
}

// ---- CLS: x_out[b,0] = x_in[b,0] + mean_t side[b*T+t] --------------------------------------------
__global__ void cls_mean_residual_kernel(const float* __restrict__ x_in, int64_t ldb_in, const float* __restrict__ side,
                                         float* __restrict__ x_out, int64_t ldb_out, int B, int T, int D) {
  const int b = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += side[((int64_t)b * T + t) * D + d];
    x_out[b * ldb_out + d] = x_in[b * ldb_in + d] + s / (float)T;
  }
}

// ---- row LayerNorm helpers (D = 768: 12 fp32 per lane, one wave per row) -------------------------
constexpr int LN_D = 768, LN_V = 3;

__device__ __forceinline__ void ln_load(const float* row, int lane, float (&v)[12]) {
#pragma unroll
  for (int i = 0; i < LN_V; ++i) {
    const float4 f = *(const float4*)(row + i * 256 + lane * 4);
    v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
  }
}
__device__ __forceinline__ void ln_load_nt(const float* row, int lane, float (&v)[12]) {  // streamed-once token rows
#pragma unroll
  for (int i = 0; i < LN_V; ++i) {
    const f32x4 f = __builtin_nontemporal_load((const f32x4*)(row + i * 256 + lane * 4));
    v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
  }
}
__device__ __forceinline__ void ln_stats(const float (&v)[12], float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += v[i];
  mean = wave_sum(s) * (1.0f / LN_D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float d = v[i] - mean;
    q += d * d;
  }
  rstd = rsqrtf(wave_sum(q) * (1.0f / LN_D) + eps);
}
__device__ __forceinline__ void ln_affine(float (&v)[12], float mean, float rstd, const float* gamma, const float* beta, int lane) {
#pragma unroll
  for (int i = 0; i < LN_V; ++i) {
    const float4 g = *(const float4*)(gamma + i * 256 + lane * 4), b = *(const float4*)(beta + i * 256 + lane * 4);
    v[4 * i] = (v[4 * i] - mean) * rstd * g.x + b.x;
    v[4 * i + 1] = (v[4 * i + 1] - mean) * rstd * g.y + b.y;
    v[4 * i + 2] = (v[4 * i + 2] - mean) * rstd * g.z + b.z;
    v[4 * i + 3] = (v[4 * i + 3] - mean) * rstd * g.w + b.w;
  }
}
template <typename T>
__device__ __forceinline__ void ln_store(T* row, int lane, const float (&v)[12]) {
#pragma unroll
  for (int i = 0; i < LN_V; ++i) {
    if constexpr (sizeof(T) == 4) {
      __builtin_nontemporal_store(f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}, (f32x4*)(row + i * 256 + lane * 4));
    } else {
      T* p = row + i * 256 + lane * 4;
      u32x2 u;
      u.x = pack2(v[4 * i], v[4 * i + 1], (T*)0);
      u.y = pack2(v[4 * i + 2], v[4 * i + 3], (T*)0);
      *(u32x2*)p = u;  // plain, not non-temporal: 154 MB at the benchmark size stay in the Infinity Cache for the GEMM that reads them next
                       // (step 174.1 -> 171.4 ms together with gather_cast; the same change on GEMM / attention outputs LOSES 5 ms)
    }
  }
}

__device__ __forceinline__ int64_t ln_src_row(int mode, int p0, int p1, int64_t m) {
  if (mode == ALPRO_MAP_IDENTITY) return m;
  if (mode == ALPRO_MAP_SKIP_CLS) return m + m / p0 + 1;
  const int T = p0, N = p1;  // FRAME_TOKENS gather
  const int64_t bt = m / (N + 1);
  const int j = (int)(m - bt * (N + 1));
  const int64_t b = bt / T;
  const int t = (int)(bt - b * T);
  const int64_t base = b * (1 + (int64_t)N * T);
  return j == 0 ? base : base + 1 + (int64_t)(j - 1) * T + t;
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, T* __restrict__ y, int64_t ldy,
                                                            float* __restrict__ y32, float* __restrict__ mean_o,
                                                            float* __restrict__ rstd_o, int64_t rows, int mode, int p0, int p1) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t m = wave; m < rows; m += nwaves) {
    float v[12], mean, rstd;
    ln_load_nt(x + ln_src_row(mode, p0, p1, m) * ldx, lane, v);
    ln_stats(v, eps, mean, rstd);
    ln_affine(v, mean, rstd, gamma, beta, lane);
    ln_store<T>(y + m * ldy, lane, v);
    if (y32) ln_store<float>(y32 + m * LN_D, lane, v);
    if (mean_o && lane == 0) {
      mean_o[m] = mean;
      rstd_o[m] = rstd;
    }
  }
}


// ---- the fusion encoder's input as a row gather (round 5) -----------------------------------------------------------------------
// alpro_models.py:278-281,325-330,360-363 build the three fusion batches with torch.cat over text / video embeddings and gathers of the
// hard negatives ("text_embeds[neg_text]"); batched into one 4B-sequence pass that was five cat kernels over up to 186 MB, an index kernel,
// and under autograd three sorted index_put_ plus the accumulation adds of every tensor that is used more than once (1.2 ms per step).
// Here sequence s of the fusion batch IS (text pool row ti[s], video pool row vi[s]): one pass writes the fp32 stream and the 16-bit
// operand copy of the first fusion layer, and the backward sums, for every pool row, the gradients of the sequences that used it in
// ascending sequence order (no atomics: bit-reproducible).
template <typename T>
__global__ __launch_bounds__(256) void gather_seq_fwd_kernel(const float* __restrict__ text, const float* __restrict__ video, const int64_t* __restrict__ ti,
                                                            const int64_t* __restrict__ vi, float* __restrict__ out32, T* __restrict__ out_t, int S, int Lt, int Lv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = Lt + Lv;
  const int64_t rows = (int64_t)S * L;
  for (int64_t m = (int64_t)blockIdx.x * 4 + wave; m < rows; m += (int64_t)gridDim.x * 4) {
    const int s = (int)(m / L), l = (int)(m - (int64_t)s * L);
    const float* src = l < Lt ? text + ((int64_t)ti[s] * Lt + l) * LN_D : video + ((int64_t)vi[s] * Lv + (l - Lt)) * LN_D;
    float v[12];
    ln_load(src, lane, v);
    ln_store<float>(out32 + m * LN_D, lane, v);
    if (out_t) ln_store<T>(out_t + m * LN_D, lane, v);
  }
}

// one wave per pool row (text rows first, then video rows): d[pool row] = sum over the sequences s (ascending) whose index points at it of
// d32[s, l] (+ d_t[s, l], the 16-bit part of the gradient the first fusion layer hands back: xbert._BertRun.backward)
template <typename T>
__global__ __launch_bounds__(256) void gather_seq_bwd_kernel(const float* __restrict__ d32, const T* __restrict__ d_t, const int64_t* __restrict__ ti,
                                                            const int64_t* __restrict__ vi, float* __restrict__ dtext, float* __restrict__ dvideo, int S, int Pt,
                                                            int Pv, int Lt, int Lv) {
  extern __shared__ int idx_s[];   // ti | vi
  for (int i = threadIdx.x; i < 2 * S; i += blockDim.x) idx_s[i] = (int)(i < S ? ti[i] : vi[i - S]);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = Lt + Lv;
  const int64_t rows = (int64_t)Pt * Lt + (int64_t)Pv * Lv;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const bool is_text = r < (int64_t)Pt * Lt;
    const int64_t q = is_text ? r : r - (int64_t)Pt * Lt;
    const int Lp = is_text ? Lt : Lv;
    const int pidx = (int)(q / Lp), l = (int)(q - (int64_t)pidx * Lp);
    const int* ix = idx_s + (is_text ? 0 : S);
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int s = 0; s < S; ++s) {
      if (ix[s] != pidx) continue;
      const int64_t m = (int64_t)s * L + (is_text ? l : Lt + l);
      float v[12];
      ln_load_nt(d32 + m * LN_D, lane, v);
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] += v[i];
      if (d_t) {
#pragma unroll
        for (int i = 0; i < LN_V; ++i) {
          const u32x2 u = *(const u32x2*)(d_t + m * LN_D + i * 256 + lane * 4);
          const uint32_t ux = u.x, uy = u.y;
          acc[4 * i] += to_f32(T{(uint16_t)(ux & 0xFFFFu)});
          acc[4 * i + 1] += to_f32(T{(uint16_t)(ux >> 16)});
          acc[4 * i + 2] += to_f32(T{(uint16_t)(uy & 0xFFFFu)});
          acc[4 * i + 3] += to_f32(T{(uint16_t)(uy >> 16)});
        }
      }
    }
    ln_store<float>((is_text ? dtext : dvideo) + q * LN_D, lane, acc);
  }
}

// ---- residual add fused into the LayerNorm that follows it ---------------------------------------------------------------------
// Every residual branch of the path ends in "x' = x + scale * Linear(...)" and is followed by a LayerNorm of x' (vit.py:162 -> :180,
// :196 -> :200; xbert.py:358-359, 436-437).  Done in the Linear's GEMM epilogue, the fp32 read-modify-write of x (8 B / element through a
// CU's ~26 GB/s epilogue path) made the two N = 768, K = 768 projections of a ViT block the slowest GEMMs of the model (0.16-0.19 of the
// MFMA peak, round 2).  Here the GEMM writes only its 16-bit output `delta` in plain row order (the 16-byte-store fast path) and THIS
// streaming kernel does the add, writes x' (fp32, the next residual / the LayerNorm-backward input) and the normalised rows: the same
// HBM bytes in total, moved at the streaming rate.  The row maps of the divided space-time block live here too, so the CLS side
// buffer and alpro_cls_mean_residual are gone:
//   ALPRO_ADD_IDENTITY      row m:  v = x_in[m] + delta[m]                                             -> x_out[m], y[m]
//   ALPRO_ADD_PRE_SPATIAL   row m = (b*T + t)*(N+1) + j of the frame-token gather (vit.py:165-172), r = its token row:
//                           j > 0: v = x_in[r] + delta[r - b - 1] + dbias   (delta = temporal branch in x[:, 1:] order, vit.py:162)
//                           j = 0: v = x_in[r]                              (the CLS token skips the temporal branch)   -> x_out[r], y[m]
//   ALPRO_ADD_PRE_MLP       row r = b*S + k of the token tensor: k > 0 (patch n, frame t): v = x_in[r] + delta[(b*T+t)*(N+1) + 1 + n];
//                           k = 0: v = x_in[r] + mean_t delta[(b*T+t)*(N+1)]  (vit.py:184-196)                             -> x_out[r], y[r]
//   ALPRO_ADD_PRE_TEMPORAL  row r = b*S + k: v = x_in[r] + delta[r] -> x_out[r];  k > 0: y[r - b - 1] = LN(v)  (MLP branch of the
//                           previous block folded into the next block's temporal LayerNorm, vit.py:212 -> :154)
// Deferred temporal add (round 6, alpro_add_layernorm_pre_mlp2): PRE_MLP with a second delta -- the temporal branch in x[:, 1:] order, added
// FIRST (then its bias, then the spatial delta: the order of fp32 additions of the PRE_SPATIAL + PRE_MLP pair, so x' is bit for bit the same).
// The inference forward then runs PRE_SPATIAL with x_out = NULL: the intermediate x + temporal branch is never written (-3 KB of 9 per row there,
// +1.5 KB here).
template <typename T>
__device__ __forceinline__ void add_delta_row(const T* drow, int lane, float (&v)[12], float w) {
#pragma unroll
  for (int i = 0; i < LN_V; ++i) {
    if constexpr (sizeof(T) == 4) {
      const f32x4 f = __builtin_nontemporal_load((const f32x4*)(drow + i * 256 + lane * 4));
      v[4 * i] += w * f.x; v[4 * i + 1] += w * f.y; v[4 * i + 2] += w * f.z; v[4 * i + 3] += w * f.w;
    } else {
      const u32x2 u = __builtin_nontemporal_load((const u32x2*)(drow + i * 256 + lane * 4));
      const uint32_t ux = u.x, uy = u.y;
      v[4 * i] += w * to_f32(T{(uint16_t)(ux & 0xFFFFu)});
      v[4 * i + 1] += w * to_f32(T{(uint16_t)(ux >> 16)});
      v[4 * i + 2] += w * to_f32(T{(uint16_t)(uy & 0xFFFFu)});
      v[4 * i + 3] += w * to_f32(T{(uint16_t)(uy >> 16)});
    }
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x_in, const T* __restrict__ delta, const float* __restrict__ dbias,
                                                                float* __restrict__ x_out, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float eps, T* __restrict__ y, float* __restrict__ y32, int64_t rows, int p0, int p1,
                                                                const T* __restrict__ delta2 = nullptr, const float* __restrict__ dbias2 = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int Tn = p0, N = p1;
  const int64_t S = 1 + (int64_t)N * Tn;
  for (int64_t m = wave; m < rows; m += nwaves) {
    int64_t r = m, yrow = m;       // token row of x_in / x_out; row of y
    const T* d0 = nullptr;         // delta row added with weight 1
    bool write_x = true, write_y = true, cls_mean = false;
    int64_t b = 0;
    if constexpr (MODE == ALPRO_ADD_IDENTITY) {
      d0 = delta + m * LN_D;
    } else if constexpr (MODE == ALPRO_ADD_PRE_SPATIAL) {
      const int64_t bt = m / (N + 1);
      const int j = (int)(m - bt * (N + 1));
      b = bt / Tn;
      const int t = (int)(bt - b * Tn);
      if (j == 0) {
        r = b * S;
        write_x = (t == 0) && (x_out != x_in);  // one writer per CLS row; in place there is nothing to write
      } else {
        r = b * S + 1 + (int64_t)(j - 1) * Tn + t;
        d0 = delta + (r - b - 1) * LN_D;
      }
    } else if constexpr (MODE == ALPRO_ADD_PRE_MLP) {
      b = m / S;
      const int64_t k = m - b * S;
      if (k == 0) {
        cls_mean = true;
      } else {
        const int64_t n = (k - 1) / Tn;
        const int t = (int)((k - 1) - n * Tn);
        d0 = delta + ((b * Tn + t) * (N + 1) + 1 + n) * LN_D;
      }
    } else {  // PRE_TEMPORAL
      b = m / S;
      const int64_t k = m - b * S;
      d0 = delta + m * LN_D;
      write_y = k != 0;
      yrow = m - b - 1;
    }
    float v[12];
    ln_load_nt(x_in + r * LN_D, lane, v);
    if constexpr (MODE == ALPRO_ADD_PRE_MLP) {
      if (delta2 && !cls_mean) {  // the deferred temporal branch of this patch row (x[:, 1:] order), its bias behind it
        add_delta_row<T>(delta2 + (r - b - 1) * LN_D, lane, v, 1.0f);
        if (dbias2) {
#pragma unroll
          for (int i = 0; i < LN_V; ++i) {
            const float4 f = *(const float4*)(dbias2 + i * 256 + lane * 4);
            v[4 * i] += f.x; v[4 * i + 1] += f.y; v[4 * i + 2] += f.z; v[4 * i + 3] += f.w;
          }
        }
      }
    }
    if (d0) {
      add_delta_row<T>(d0, lane, v, 1.0f);
      if (dbias) {
#pragma unroll
        for (int i = 0; i < LN_V; ++i) {
          const float4 f = *(const float4*)(dbias + i * 256 + lane * 4);
          v[4 * i] += f.x; v[4 * i + 1] += f.y; v[4 * i + 2] += f.z; v[4 * i + 3] += f.w;
        }
      }
    }
    if (MODE == ALPRO_ADD_PRE_MLP && cls_mean) {  // frame mean of the T CLS rows of the spatial branch (vit.py:187)
      float a[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) a[i] = 0.f;
      for (int t = 0; t < Tn; ++t) add_delta_row<T>(delta + ((b * Tn + t) * (N + 1)) * LN_D, lane, a, 1.0f);
      const float inv = 1.0f / (float)Tn;
#pragma unroll
      for (int i = 0; i < 12; ++i) v[i] += a[i] * inv;
    }
    if (x_out && write_x) ln_store<float>(x_out + r * LN_D, lane, v);
    if (!write_y) continue;
    float mean, rstd;
    ln_stats(v, eps, mean, rstd);
    ln_affine(v, mean, rstd, gamma, beta, lane);
    ln_store<T>(y + yrow * LN_D, lane, v);
    if constexpr (sizeof(T) != 4) {
      if (y32) ln_store<float>(y32 + yrow * LN_D, lane, v);
    }
  }
}

// final norm + temporal mean pool: out row (b, 0) = LN(x[b,0]); (b, 1+n) = mean_t LN(x[b, 1+n*T+t])
template <typename T>
__global__ __launch_bounds__(256) void vit_final_pool_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float* __restrict__ out32,
                                                             T* __restrict__ out_t, int B, int Tn, int N) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = (int64_t)B * (N + 1);
  if (wave >= total) return;
  const int64_t b = wave / (N + 1);
  const int j = (int)(wave - b * (N + 1));
  const float* base = x + b * (1 + (int64_t)N * Tn) * LN_D;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  const int cnt = j == 0 ? 1 : Tn;
  for (int t = 0; t < cnt; ++t) {
    const float* row = j == 0 ? base : base + (1 + (int64_t)(j - 1) * Tn + t) * LN_D;
    float v[12], mean, rstd;
    ln_load(row, lane, v);
    ln_stats(v, eps, mean, rstd);
    ln_affine(v, mean, rstd, gamma, beta, lane);
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] += v[i];
  }
  const float inv = 1.0f / (float)cnt;
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] *= inv;
  ln_store<float>(out32 + wave * LN_D, lane, acc);
  if (out_t) ln_store<T>(out_t + wave * LN_D, lane, acc);
}

// BERT embeddings: word[id] + type0 + pos[l] -> LN
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type0,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         float* __restrict__ y32, T* __restrict__ y_t, float* __restrict__ mean_o,
                                                         float* __restrict__ rstd_o, int rows, int L, float drop_p, uint32_t drop_seed) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int l = m % L;
  float v[12], w[12], mean, rstd;
  ln_load(word + ids[m] * LN_D, lane, v);
  ln_load(type0, lane, w);
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] += w[i];
  ln_load(pos + (int64_t)l * LN_D, lane, w);
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] += w[i];
  ln_stats(v, eps, mean, rstd);
  ln_affine(v, mean, rstd, gamma, beta, lane);
  if (drop_seed) {
    const uint32_t th = drop_thresh24(drop_p);
    const float ks = 1.0f / (1.0f - drop_p);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const uint64_t idx = (uint64_t)m * LN_D + (uint64_t)((i >> 2) * 256 + lane * 4 + (i & 3));
      v[i] = drop_keep(drop_seed, idx, th) ? v[i] * ks : 0.f;
    }
  }
  ln_store<float>(y32 + (int64_t)m * LN_D, lane, v);
  if (y_t) ln_store<T>(y_t + (int64_t)m * LN_D, lane, v);
  if (mean_o && lane == 0) {
    mean_o[m] = mean;
    rstd_o[m] = rstd;
  }
}

inline int grid_for(int64_t work_items, int per_block, int cap = 256 * 16) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace
}  // namespace alpro

using namespace alpro;

extern "C" const char* alpro_hip_last_error(void) { return g_err; }
extern "C" int alpro_hip_abi_version(void) { return ALPRO_HIP_ABI_VERSION; }
extern "C" int alpro_hip_set_option(const char* name, int value) {
  using namespace alpro;
  for (int i = 0; name && i < OPT_COUNT; ++i)
    if (!strcmp(name, kOptNames[i])) {
      if (!option_allowed(i, value)) {
        set_error("alpro_hip_set_option: %s=%d is an ablation / measurement-only variant, only available in the measurement build (-DALPRO_ABLATIONS)", name, value);
        return ALPRO_ERR_INVALID;
      }
      __atomic_store_n(&g_opts[i], value, __ATOMIC_RELAXED);
      return ALPRO_OK;
    }
  set_error("alpro_hip_set_option: unknown option '%s'", name ? name : "(null)");
  return ALPRO_ERR_INVALID;
}

extern "C" int alpro_hip_set_stream_option(void* stream, const char* name, int value) {
  using namespace alpro;
  int which = -1;
  for (int i = 0; name && i < OPT_COUNT; ++i)
    if (!strcmp(name, kOptNames[i])) which = i;
  if (which < 0) {
    set_error("alpro_hip_set_stream_option: unknown option '%s'", name ? name : "(null)");
    return ALPRO_ERR_INVALID;
  }
  if (value >= 0 && !option_allowed(which, value)) {
    set_error("alpro_hip_set_stream_option: %s=%d is an ablation / measurement-only variant, only available in the measurement build (-DALPRO_ABLATIONS)", name, value);
    return ALPRO_ERR_INVALID;
  }
  SpinGuard g;
  int at = -1;
  for (int i = 0; i < g_nsopts; ++i)
    if (g_sopts[i].st == (hipStream_t)stream && g_sopts[i].which == which) at = i;
  if (value < 0) {   // clear the override
    if (at >= 0) {
      g_sopts[at] = g_sopts[g_nsopts - 1];
      __atomic_store_n(&g_nsopts, g_nsopts - 1, __ATOMIC_RELAXED);
    }
    return ALPRO_OK;
  }
  if (at < 0) {
    if (g_nsopts == kMaxStreamOpts) {
      set_error("alpro_hip_set_stream_option: more than %d (stream, option) overrides", kMaxStreamOpts);
      return ALPRO_ERR_INVALID;
    }
    at = g_nsopts;
    g_sopts[at].st = (hipStream_t)stream;
    g_sopts[at].which = which;
    g_sopts[at].value = value;
    __atomic_store_n(&g_nsopts, g_nsopts + 1, __ATOMIC_RELAXED);
  }
  g_sopts[at].value = value;
  return ALPRO_OK;
}

extern "C" size_t alpro_hip_sched_workspace_bytes(void) { return (size_t)2 * alpro::SCHED_BLOCK_U32 * sizeof(uint32_t); }

extern "C" int alpro_hip_set_sched_workspace(void* stream, void* ptr, size_t bytes) { return alpro::sched_set_workspace((hipStream_t)stream, ptr, bytes); }

extern "C" int alpro_hip_release_stream(void* stream) {
  using namespace alpro;
  {   // every per-stream option override of that stream
    SpinGuard g;
    for (int i = 0; i < g_nsopts;) {
      if (g_sopts[i].st == (hipStream_t)stream) {
        g_sopts[i] = g_sopts[g_nsopts - 1];
        __atomic_store_n(&g_nsopts, g_nsopts - 1, __ATOMIC_RELAXED);
      } else {
        ++i;
      }
    }
  }
  return sched_set_workspace((hipStream_t)stream, nullptr, 0);   // ... and its tile-scheduler slot (a caller-owned workspace is simply forgotten)
}

extern "C" int alpro_cast_from_f32(const float* src, void* dst, int dtype, int64_t n, void* stream) {
  ALPRO_CHECK(src && dst && n >= 0, "alpro_cast_from_f32: bad args");
  if (n == 0) return ALPRO_OK;
  ALPRO_CHECK(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "alpro_cast_from_f32: pointers must be 16-byte aligned");
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(cast_kernel<T>, dim3(grid_for(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, n));
  return check_launch("alpro_cast_from_f32");
}

namespace alpro {
namespace {
// ---- clip preparation: ImageNorm + the MPM random-erase crop in ONE pass (data_utils.py:437-457, dataset_pretrain_sparse.py:277-311,
// dataloader.py:104-115).  raw (B, T, 3, H, W) pixels (uint8 or float) are read once; per element
//   n(x) = (x * scale - mean[c]) / std[c]        (torch's img.div_(255.) is a multiply by 1/255 on the device; sub_/div_ by the tensors are exact)
//   visual = n(x);  crop = n(inside ? x : 0);  context = n(inside ? 0 : x)     inside = the sample's patch-aligned rectangle
// The erase happens on RAW pixels, so erased regions hold n(0) = -mean/std, not 0 -- exactly what the reference feeds the encoders.
template <typename In>
__global__ __launch_bounds__(256) void prepare_clips_kernel(const In* __restrict__ raw, const int* __restrict__ boxes, float scale, float m0, float m1,
                                                            float m2, float s0, float s1, float s2, float* __restrict__ vis, float* __restrict__ crop,
                                                            float* __restrict__ ctx, int T, int H, int W, int64_t nvec) {
  const int W4 = W >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W4) * 4;
    int64_t r = i / W4;
    const int h = (int)(r % H);
    r /= H;
    const int c = (int)(r % 3);
    const int b = (int)(r / 3 / T);
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    float x[4];
    if constexpr (sizeof(In) == 1) {
      const uint32_t u = *(const uint32_t*)(raw + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (float)((u >> (8 * e)) & 0xffu);
    } else {
      const f32x4 v = __builtin_nontemporal_load((const f32x4*)(raw + i * 4));
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    }
    float nv[4], nc[4], nx[4];
    const float zero_n = (0.f * scale - mean) / sd;
    bool row_in = false;
    int left = 0, right = 0;
    if (boxes) {
      const int top = boxes[4 * b], lf = boxes[4 * b + 1], bh = boxes[4 * b + 2], bw = boxes[4 * b + 3];
      row_in = h >= top && h < top + bh;
      left = lf;
      right = lf + bw;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      nv[e] = (x[e] * scale - mean) / sd;
      const bool inside = row_in && (w + e) >= left && (w + e) < right;
      nc[e] = inside ? nv[e] : zero_n;
      nx[e] = inside ? zero_n : nv[e];
    }
    __builtin_nontemporal_store(f32x4{nv[0], nv[1], nv[2], nv[3]}, (f32x4*)(vis + i * 4));
    if (crop) __builtin_nontemporal_store(f32x4{nc[0], nc[1], nc[2], nc[3]}, (f32x4*)(crop + i * 4));
    if (ctx) __builtin_nontemporal_store(f32x4{nx[0], nx[1], nx[2], nx[3]}, (f32x4*)(ctx + i * 4));
  }
}
}  // namespace
}  // namespace alpro

extern "C" int alpro_prepare_clips(const void* raw, int raw_is_u8, const int* boxes, float scale, const float* mean3, const float* std3, float* visual,
                                   float* crop, float* context, int B, int T, int H, int W, void* stream) {
  using namespace alpro;
  ALPRO_CHECK(raw && mean3 && std3 && visual && B > 0 && T > 0 && H > 0 && W > 0, "alpro_prepare_clips: bad args");
  ALPRO_CHECK(W % 4 == 0, "alpro_prepare_clips: width %d must be a multiple of 4", W);
  ALPRO_CHECK((crop == nullptr && context == nullptr) || boxes, "alpro_prepare_clips: crop / context outputs need the erase boxes");
  const int64_t nvec = (int64_t)B * T * 3 * H * (W / 4);
  const unsigned grid = (unsigned)((nvec + 255) / 256 < 256 * 16 ? (nvec + 255) / 256 : 256 * 16);
  if (raw_is_u8)
    hipLaunchKernelGGL(prepare_clips_kernel<uint8_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)raw, boxes, scale, mean3[0], mean3[1], mean3[2],
                       std3[0], std3[1], std3[2], visual, crop, context, T, H, W, nvec);
  else
    hipLaunchKernelGGL(prepare_clips_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)raw, boxes, scale, mean3[0], mean3[1], mean3[2],
                       std3[0], std3[1], std3[2], visual, crop, context, T, H, W, nvec);
  return check_launch("alpro_prepare_clips");
}

extern "C" int alpro_patchify(const float* img, void* out, int dtype, int BT, int C, int Himg, int Wimg, void* stream) {
  ALPRO_CHECK(img && out && BT > 0 && C > 0, "alpro_patchify: bad args");
  ALPRO_CHECK(Himg % 16 == 0 && Wimg % 16 == 0, "alpro_patchify: image %dx%d not a multiple of the 16x16 patch", Himg, Wimg);
  const int64_t total = (int64_t)BT * (Himg / 16) * (Wimg / 16) * C * 256 / (dtype == ALPRO_F32 ? 4 : 8);
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(patchify_kernel<T>, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, img, (T*)out, BT, C, Himg, Wimg));
  return check_launch("alpro_patchify");
}

extern "C" int alpro_gather_seq_fwd(const float* text, const float* video, const int64_t* ti, const int64_t* vi, float* out32, void* out_t, int dtype, int S,
                                    int Lt, int Lv, int D, void* stream) {
  ALPRO_CHECK(text && video && ti && vi && out32 && S > 0 && Lt > 0 && Lv > 0, "alpro_gather_seq_fwd: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_gather_seq_fwd: D=%d unsupported (hidden size is 768 on this path)", D);
  ALPRO_CHECK(dtype == ALPRO_BF16 || dtype == ALPRO_F16 || !out_t, "alpro_gather_seq_fwd: the operand copy is 16-bit (fp32 mode: pass NULL)");
  const int grid = grid_for((int64_t)S * (Lt + Lv), 4, 256 * 32);
  if (dtype == ALPRO_BF16) hipLaunchKernelGGL(gather_seq_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, text, video, ti, vi, out32, (bf16_t*)out_t, S, Lt, Lv);
  else hipLaunchKernelGGL(gather_seq_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, text, video, ti, vi, out32, (f16_t*)out_t, S, Lt, Lv);
  return check_launch("alpro_gather_seq_fwd");
}

extern "C" int alpro_gather_seq_bwd(const float* d32, const void* d_t, int dtype, const int64_t* ti, const int64_t* vi, float* dtext, float* dvideo, int S, int Pt,
                                    int Pv, int Lt, int Lv, int D, void* stream) {
  ALPRO_CHECK(d32 && ti && vi && dtext && dvideo && S > 0 && S <= 8192 && Pt > 0 && Pv > 0 && Lt > 0 && Lv > 0, "alpro_gather_seq_bwd: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_gather_seq_bwd: D=%d unsupported (hidden size is 768 on this path)", D);
  ALPRO_CHECK(dtype == ALPRO_BF16 || dtype == ALPRO_F16 || !d_t, "alpro_gather_seq_bwd: the 16-bit gradient part needs a 16-bit dtype");
  const int grid = grid_for((int64_t)Pt * Lt + (int64_t)Pv * Lv, 4, 256 * 32);
  const size_t lds = (size_t)2 * S * sizeof(int);
  if (dtype == ALPRO_BF16) hipLaunchKernelGGL(gather_seq_bwd_kernel<bf16_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, d32, (const bf16_t*)d_t, ti, vi, dtext, dvideo, S, Pt, Pv, Lt, Lv);
  else hipLaunchKernelGGL(gather_seq_bwd_kernel<f16_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, d32, (const f16_t*)d_t, ti, vi, dtext, dvideo, S, Pt, Pv, Lt, Lv);
  return check_launch("alpro_gather_seq_bwd");
}

extern "C" int alpro_cls_mean_residual(const float* x_in, int64_t ld_batch_in, const float* side, float* x_out,
                                       int64_t ld_batch_out, int B, int T, int D, void* stream) {
  ALPRO_CHECK(x_in && side && x_out && B > 0 && T > 0 && D > 0, "alpro_cls_mean_residual: bad args");
  hipLaunchKernelGGL(cls_mean_residual_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x_in, ld_batch_in, side, x_out, ld_batch_out, B, T, D);
  return check_launch("alpro_cls_mean_residual");
}

extern "C" int alpro_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* y,
                                   int y_dtype, int64_t ldy, float* y32, float* mean, float* rstd, int rows, int D,
                                   int map_mode, int map_p0, int map_p1, void* stream) {
  ALPRO_CHECK(x && gamma && beta && y && rows > 0, "alpro_layernorm_fwd: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_layernorm_fwd: D=%d unsupported (hidden size is 768 on this path)", D);
  ALPRO_CHECK(map_mode >= 0 && map_mode <= ALPRO_MAP_FRAME_TOKENS, "alpro_layernorm_fwd: bad map_mode %d", map_mode);
  ALPRO_CHECK((mean == nullptr) == (rstd == nullptr), "alpro_layernorm_fwd: mean and rstd go together");
  ALPRO_CHECK(ldx % 4 == 0 && ldy % 4 == 0, "alpro_layernorm_fwd: row strides must be multiples of 4");
  ALPRO_DISPATCH_DTYPE(y_dtype, T, hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3(grid_for(rows, 4, 256 * 32)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, eps, (T*)y, ldy, y32, mean, rstd, (int64_t)rows, map_mode, map_p0, map_p1));
  return check_launch("alpro_layernorm_fwd");
}

namespace {
template <typename T>
int launch_add_ln(int mode, const float* x_in, const void* delta, const float* dbias, float* x_out, const float* gamma, const float* beta, float eps, void* y,
                  float* y32, int64_t rows, int p0, int p1, hipStream_t st) {
  const dim3 grid(grid_for(rows, 4, 256 * 32)), blk(256);
#define ALPRO_ADD_LN_CASE(M) \
  case M: hipLaunchKernelGGL((add_layernorm_fwd_kernel<T, M>), grid, blk, 0, st, x_in, (const T*)delta, dbias, x_out, gamma, beta, eps, (T*)y, y32, rows, p0, p1); break;
  switch (mode) {
    ALPRO_ADD_LN_CASE(ALPRO_ADD_IDENTITY)
    ALPRO_ADD_LN_CASE(ALPRO_ADD_PRE_SPATIAL)
    ALPRO_ADD_LN_CASE(ALPRO_ADD_PRE_MLP)
    ALPRO_ADD_LN_CASE(ALPRO_ADD_PRE_TEMPORAL)
  }
#undef ALPRO_ADD_LN_CASE
  return check_launch("alpro_add_layernorm_fwd");
}
}  // namespace

extern "C" int alpro_add_layernorm_fwd(const float* x_in, const void* delta, int dtype, const float* delta_bias, int add_mode, float* x_out,
                                       const float* gamma, const float* beta, float eps, void* y, float* y32, int64_t rows, int D, int p0, int p1,
                                       void* stream) {
  ALPRO_CHECK(x_in && delta && gamma && beta && y && rows > 0, "alpro_add_layernorm_fwd: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_add_layernorm_fwd: D=%d unsupported (hidden size is 768 on this path)", D);
  ALPRO_CHECK(add_mode >= ALPRO_ADD_IDENTITY && add_mode <= ALPRO_ADD_PRE_TEMPORAL, "alpro_add_layernorm_fwd: bad add_mode %d", add_mode);
  ALPRO_CHECK(add_mode == ALPRO_ADD_IDENTITY || (p0 > 0 && p1 > 0), "alpro_add_layernorm_fwd: the divided space-time modes need p0 = T, p1 = N");
  ALPRO_CHECK(add_mode == ALPRO_ADD_IDENTITY || rows % (add_mode == ALPRO_ADD_PRE_SPATIAL ? (int64_t)p0 * (p1 + 1) : 1 + (int64_t)p0 * p1) == 0,
              "alpro_add_layernorm_fwd: rows=%lld is not a whole number of clips", (long long)rows);
  ALPRO_CHECK(!(dtype == ALPRO_F32 && y32), "alpro_add_layernorm_fwd: y already is fp32 in exact mode");
  ALPRO_CHECK(((uintptr_t)x_in % 16) == 0 && ((uintptr_t)delta % 16) == 0 && ((uintptr_t)y % 16) == 0, "alpro_add_layernorm_fwd: pointers must be 16-byte aligned");
  ALPRO_DISPATCH_DTYPE(dtype, T, return launch_add_ln<T>(add_mode, x_in, delta, delta_bias, x_out, gamma, beta, eps, y, y32, rows, p0, p1, (hipStream_t)stream));
  return ALPRO_OK;
}

extern "C" int alpro_add_layernorm_pre_mlp2(const float* x_in, const void* delta_t, const float* delta_t_bias, const void* delta_s, int dtype, float* x_out,
                                            const float* gamma, const float* beta, float eps, void* y, int64_t rows, int D, int T, int N, void* stream) {
  ALPRO_CHECK(x_in && delta_t && delta_s && gamma && beta && y && rows > 0, "alpro_add_layernorm_pre_mlp2: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_add_layernorm_pre_mlp2: D=%d unsupported (hidden size is 768 on this path)", D);
  ALPRO_CHECK(T > 0 && N > 0 && rows % (1 + (int64_t)T * N) == 0, "alpro_add_layernorm_pre_mlp2: rows=%lld is not a whole number of clips of 1 + %d x %d tokens", (long long)rows, N, T);
  ALPRO_CHECK(((uintptr_t)x_in % 16) == 0 && ((uintptr_t)delta_t % 16) == 0 && ((uintptr_t)delta_s % 16) == 0 && ((uintptr_t)y % 16) == 0,
              "alpro_add_layernorm_pre_mlp2: pointers must be 16-byte aligned");
  const dim3 grid(grid_for(rows, 4, 256 * 32)), blk(256);
  ALPRO_DISPATCH_DTYPE(dtype, T_, hipLaunchKernelGGL((add_layernorm_fwd_kernel<T_, ALPRO_ADD_PRE_MLP>), grid, blk, 0, (hipStream_t)stream, x_in, (const T_*)delta_s,
                                                     (const float*)nullptr, x_out, gamma, beta, eps, (T_*)y, (float*)nullptr, rows, T, N, (const T_*)delta_t, delta_t_bias));
  return check_launch("alpro_add_layernorm_pre_mlp2");
}

extern "C" int alpro_vit_final_pool(const float* x, const float* gamma, const float* beta, float eps, float* out32, void* out_t,
                                    int dtype, int B, int T, int N, int D, void* stream) {
  ALPRO_CHECK(x && gamma && beta && out32 && B > 0 && T > 0 && N > 0, "alpro_vit_final_pool: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_vit_final_pool: D=%d unsupported", D);
  const int64_t total = (int64_t)B * (N + 1);
  ALPRO_DISPATCH_DTYPE(dtype, T_, hipLaunchKernelGGL(vit_final_pool_kernel<T_>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, out32, (T_*)out_t, B, T, N));
  return check_launch("alpro_vit_final_pool");
}

extern "C" int alpro_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                                    const float* gamma, const float* beta, float eps, float* y32, void* y_t, int dtype,
                                    float* mean, float* rstd, int rows, int L, int D, float drop_p, uint32_t drop_seed, void* stream) {
  ALPRO_CHECK(ids && word && pos && type0 && gamma && beta && y32 && rows > 0 && L > 0, "alpro_bert_embed_fwd: bad args");
  ALPRO_CHECK(D == LN_D, "alpro_bert_embed_fwd: D=%d unsupported", D);
  ALPRO_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(bert_embed_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, word, pos, type0, gamma, beta, eps, y32, (T*)y_t, mean, rstd, rows, L, drop_p, drop_seed));
  return check_launch("alpro_bert_embed_fwd");
}
