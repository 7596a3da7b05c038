#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned a = 100 + l, b = 200 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0];
  out[64 + l] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
