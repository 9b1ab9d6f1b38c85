// Hardware probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950.
// Prints, for a linear per-lane address (lane l -> elements 4l..4l+3), which LDS element index
// each (lane, j) of the result came from.   hipcc --offload-arch=gfx950 tools/probe_tr.hip -o probe_tr
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * stride_elems));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}
int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {4, 16, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
    uint16_t h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d elems per lane (lane: 4 source element indices)\n", stride);
    for (int l = 0; l < 64; ++l) printf("L%02d:%5d%5d%5d%5d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "  |");
  }
  return 0;
}
