// A stand-in for a collective's kernels on ONE GPU: `n` workgroups of 256 threads with 64 KiB of LDS each -- the footprint of
// an RCCL channel's kernel -- that stay resident for `ticks` ticks of the 100 MHz wall clock, moving a trickle of memory traffic meanwhile.
// While they run, a persistent one-workgroup-per-CU GEMM (160 KiB of LDS) finds `n` CUs taken.  Built at run time by tools/overlap_contention.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void cu_thief_kernel(float* buf, uint64_t ticks, int* started) {
  extern __shared__ float lds[];
  const uint64_t t0 = wall_clock64();
  if (threadIdx.x == 0) atomicAdd(started, 1);
  float acc = 0.f;
  float* mine = buf + (size_t)blockIdx.x * 65536;
  int i = threadIdx.x;
  while (wall_clock64() - t0 < ticks) {
    lds[i & 16383] = acc;
    acc += mine[i & 65535];
    mine[(i + 32768) & 65535] = acc * 0.5f;
    i += 256;
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 123.456f) buf[0] = lds[0];
}

extern "C" int cu_thief_launch(float* buf, int n, uint64_t ticks, int* started, void* stream) {
  (void)hipFuncSetAttribute((const void*)cu_thief_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(cu_thief_kernel, dim3(n), dim3(256), 65536, (hipStream_t)stream, buf, ticks, started);
  return (int)hipGetLastError();
}
