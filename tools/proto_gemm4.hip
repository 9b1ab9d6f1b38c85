// Round-3 prototype (NOT part of the library): is a 4-wave schedule of the 256 x 256 x 64 NT tile -- one wave per SIMD, 128 x 128 per wave,
// 256 accumulator registers in AGPRs, half the LDS fragment reads per MFMA of the shipped 8-wave kernel -- a faster K loop on gfx950?
// hipcc allocates it without spills (62 VGPR + 256 AGPR).  K loop only: every lane folds its accumulators into one float at the end, so the
// number to compare is the shipped kernel's K loop with its epilogue compiled out (~1170 TF/s, profiles/r2_gemm_epilogue_experiments.txt).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto_gemm4.hip -o /tmp/proto_gemm4 && /tmp/proto_gemm4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ void dma16(const void* src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
template <int PERSIST>
__global__ __launch_bounds__(256, 1) void g4(const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ out, int64_t lda_b, int64_t ldw_b, int nk,
                                             int ntn, int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  int a_row[4], b_row[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a_row[i] = wr * 128 + i * 32 + (lane & 31); b_row[i] = wc * 128 + i * 32 + (lane & 31); }
  const int khalf = lane >> 5;
  float fold = 0.f;
  for (int tile = blockIdx.x; tile < nblk; tile += gridDim.x) {
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const char* a_src[8]; const char* w_src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wave + 4 * i) * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      a_src[i] = A + (int64_t)(tm * 256 + row) * lda_b + ch * 16;
      w_src[i] = W + (int64_t)(tn * 256 + row) * ldw_b + ch * 16;
    }
    auto piece = [&](int c, int kt, int buf) {  // c = 0..15: (A, W) x 8 pieces of 1 KiB
      const int i = c >> 1;
      const char* src = ((c & 1) ? w_src[i] : a_src[i]) + (int64_t)kt * 128;
      dma16(src, __builtin_amdgcn_readfirstlane(lds_base + buf * 65536 + (c & 1) * 32768 + (wave + 4 * i) * 1024));
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __builtin_amdgcn_s_barrier();  // previous tile's readers are done with both stages
#pragma unroll
    for (int c = 0; c < 16; ++c) piece(c, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = kt + 1 < nk;
      const char* cA = smem + cur * 65536;
      const char* cW = cA + 32768;
      u32x4 fa[2][4], fb[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[0][i] = *(const u32x4*)(cA + lds_off(a_row[i], khalf)); fb[0][i] = *(const u32x4*)(cW + lds_off(b_row[i], khalf)); }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            fa[(s + 1) & 1][i] = *(const u32x4*)(cA + lds_off(a_row[i], 2 * (s + 1) + khalf));
            fb[(s + 1) & 1][i] = *(const u32x4*)(cW + lds_off(b_row[i], 2 * (s + 1) + khalf));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[s & 1][i]), __builtin_bit_cast(bf16x8_t, fb[s & 1][j]), acc[i][j], 0, 0, 0);
            const int q = s * 16 + i * 4 + j;             // 0..63: one DMA piece of the next K-tile every 4th MFMA, all out by MFMA 61
            if (more && (q & 3) == 1) piece(q >> 2, kt + 1, cur ^ 1);
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) fold += acc[i][j][r];
    if (!PERSIST) break;
  }
  out[(int64_t)blockIdx.x * 256 + tid] = fold;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
int main() {
  const int shapes[][3] = {{100352, 2304, 768}, {100352, 768, 768}, {100352, 768, 3072}, {50176, 2304, 768}, {8192, 8192, 8192}};
  CK(hipFuncSetAttribute((const void*)g4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)g4<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; const float f = ((st >> 8) & 0xffff) / 65536.0f - 0.5f; const uint32_t u = __builtin_bit_cast(uint32_t, f); return (uint16_t)(u >> 16); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hw) v = rnd();
    char *dA, *dW; float* dOut;
    CK(hipMalloc(&dA, ha.size() * 2)); CK(hipMalloc(&dW, hw.size() * 2)); CK(hipMalloc(&dOut, 4096 * 256 * 4 * 16));
    CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const int ntn = N / 256, nblk = (M / 256) * ntn, nk = K / 64;
    for (int persist = 0; persist < 2; ++persist) {
      const int grid = persist ? (nblk < 256 ? nblk : 256) : nblk;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto launch = [&]() {
        if (persist) hipLaunchKernelGGL(g4<1>, dim3(grid), dim3(256), 131072, 0, dA, dW, dOut, (int64_t)K * 2, (int64_t)K * 2, nk, ntn, nblk);
        else hipLaunchKernelGGL(g4<0>, dim3(grid), dim3(256), 131072, 0, dA, dW, dOut, (int64_t)K * 2, (int64_t)K * 2, nk, ntn, nblk);
      };
      for (int i = 0; i < 3; ++i) launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("4-wave K loop only  M=%d N=%d K=%d  %s grid %d: %.3f ms  %.0f TF/s\n", M, N, K, persist ? "persistent" : "one tile per WG", grid, ms, 2.0 * M * N * K / ms / 1e9);
    }
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dOut));
  }
  return 0;
}
