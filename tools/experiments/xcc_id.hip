// Experiment: what does HW_REG_XCC_ID read per workgroup, and how does it relate to blockIdx % 8?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  const int n = 4096;
  unsigned* d; (void)hipMalloc(&d, n * 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
  std::vector<unsigned> h(n);
  (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; i++) printf("block %d: raw=0x%x low4=%u\n", i, h[i], h[i] & 15u);
  int hist[16][8] = {};
  for (int i = 0; i < n; i++) hist[h[i] & 15u][i % 8]++;
  for (int x = 0; x < 16; x++) { printf("xcc %2d:", x); for (int r = 0; r < 8; r++) printf(" %5d", hist[x][r]); printf("\n"); }
  return 0;
}
