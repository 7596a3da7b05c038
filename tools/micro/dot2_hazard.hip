// Is an inline-asm consumer of a v_dot2c_f32_bf16 result safe?  K independent VALU instructions sit between the dot
// product and `v_cvt_rpi_i32_f32` (inline asm, as round 1's bwd_q had it); the result is compared with the plain-C path.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ unsigned pk(float lo, float hi) { bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi; return __builtin_bit_cast(unsigned, v); }
template <int K>
__global__ void k(int* out, const float* in, int n) {
  const int l = threadIdx.x;
  int acc = 0, ref = 0;
  float filler = in[l];
  for (int it = 0; it < n; ++it) {
    const float g = in[(l + it) & 63] * 1000.f, gb = in[(l + it + 7) & 63] * 1000.f;
    const unsigned pr = pk(g, gb), w = pk(0.25f + 0.001f * it, 0.5f);
    float h = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pr), __builtin_bit_cast(bf16x2, w), 0.f, false);
    if (K >= 1) filler = filler * 1.0001f + 1.f;
    if (K >= 2) filler = filler * 1.0002f + 2.f;
    if (K >= 3) filler = filler * 1.0003f + 3.f;
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(h));
    acc += r;
    ref += (int)floorf(h + 0.5f);
  }
  out[l] = acc - ref;
  out[64 + l] = (int)filler;
}
template <int K> void run(int* d, float* in) {
  k<K><<<1, 64>>>(d, in, 64);
  int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) bad += h[i] != 0;
  printf("K=%d independent instructions between dot2c and the asm cvt: %d of 64 lanes differ\n", K, bad);
}
int main() {
  int* d; float* in; hipMalloc(&d, 128 * 4); hipMalloc(&in, 64 * 4);
  float hin[64]; for (int i = 0; i < 64; ++i) hin[i] = 0.37f * (i - 30.5f);
  hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
  run<0>(d, in); run<1>(d, in); run<2>(d, in); run<3>(d, in);
  return 0;
}
