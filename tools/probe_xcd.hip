// Which XCD does workgroup b of a 1-D grid land on?  Prints XCC_ID (hwreg 20) per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = hw; }
  // keep the CU busy a little so that blocks spread
  volatile float f = 1.f;
  for (int i = 0; i < spin; ++i) f = f * 1.0001f + 0.1f;
}
int main() {
  const int nb = 600;
  unsigned* d; hipMalloc(&d, nb * 8);
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL(k, dim3(nb), dim3(threads), 0, 0, d, 20000);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    int mism = 0;
    for (int b = 0; b < nb; ++b) if ((h[2 * b] & 0xf) != (unsigned)(b % 8)) ++mism;
    printf("threads %d: mismatches vs b%%8: %d of %d\n", threads, mism, nb);
    for (int b = 0; b < 40; ++b) printf("%u ", h[2 * b] & 0xf);
    printf("\n");
    for (int b = 256; b < 296; ++b) printf("%u ", h[2 * b] & 0xf);
    printf("\n");
  }
  return 0;
}
