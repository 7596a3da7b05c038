// v_dot2c_f32_bf16 / v_cvt_pk_bf16_f32 / wave_shr DPP semantics on gfx950 at several magnitudes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ unsigned pk(float lo, float hi) { bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi; return __builtin_bit_cast(unsigned, v); }
__global__ void k(float* out, float scale) {
  const int l = threadIdx.x;
  const float g = scale * (float)(l + 1);
  const float gb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, g), 0x138, 0xf, 0xf, true));
  const unsigned pr = pk(g, gb), w = pk(0.25f, 0.5f);
  const float h = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pr), __builtin_bit_cast(bf16x2, w), 0.f, false);
  out[l] = h;               // expect 0.25 g + 0.5 gb = scale (0.25 (l+1) + 0.5 l)
  out[64 + l] = gb;
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  for (float scale : {1.0f, 1e3f, 1e6f, 1e8f}) {
    k<<<1, 64>>>(d, scale);
    float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("scale %g: ", scale);
    for (int l : {0, 1, 2, 31, 32, 33, 63}) printf("[%d] h=%g want=%g gb=%g | ", l, h[l], scale * (0.25 * (l + 1) + 0.5 * l), h[64 + l]);
    printf("\n");
  }
  return 0;
}
