// Key preparation of the tap kernels (attn_tap.h): per key the clamped rpe-table coordinates and the sampling position
// in feature pixels (TapRec), per 32-key tile the box of its live keys (StepBox).  One wave per 64 keys, once per call.
#include "attn_tap.h"

namespace {

__device__ __forceinline__ int hmin_i(int v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = min(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ int hmax_i(int v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = max(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float hmin_f(float v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = fminf(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float hmax_f(float v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}

__global__ __launch_bounds__(256) void attn_tap_prep_kernel(bevr_attn_desc d, const float* __restrict__ key_a,
                                                            const float* __restrict__ key_b, const float* __restrict__ key_y,
                                                            const float* __restrict__ key_x, TapRec* __restrict__ rec_out,
                                                            StepBox* __restrict__ box_out, int n_wave_total) {
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);   // global wave = prob * n_step + step
  if (gw >= n_wave_total) return;
  const int lane = threadIdx.x & 63;
  const int n_step = d.Np / KT;
  const int prob = gw / n_step, step = gw % n_step;
  const size_t idx = (size_t)prob * d.Np + (size_t)step * KT + lane;
  const bool live = step * KT + lane < d.N;
  float a = key_a[idx], b = key_b[idx];
  // the same clamps as attn_keyprep.hip: every tap of a clamped key lies inside the zero-padded table
  const float aL = -(float)(d.Sp + 1), aU = (float)(d.Ht + 1);
  const float half = (float)(d.Wt / 2);
  const float bL = -(half + 2.0f), bU = (float)(d.Wt + 1);
  a = fminf(fmaxf(a, aL), aU);
  b = fminf(fmaxf(b, bL), bU);
  const int A = (int)floorf(a);
  const int amin = hmin_i(live ? A : 0x7fffffff), amax = hmax_i(live ? A : (int)0x80000000);
  const float bmin = hmin_f(live ? b : 3.0e38f), bmax = hmax_f(live ? b : -3.0e38f);
  TapRec r;
  r.a = live ? a : (amax >= amin ? (float)amin : 0.f);
  r.b = live ? b : (amax >= amin ? bmin : 0.f);
  // NaN positions sample nothing (every hat() of a NaN is 0 through fmaxf)
  r.ys = live ? key_y[idx] : TAP_YS_DEAD;
  r.xs = live ? key_x[idx] : 0.f;
  // the tap contract (attn_tap.h): a live key samples inside feature rows 0..3 x columns 0..2.  A key beyond them would
  // silently lose the taps the 12-pixel grid does not hold; the bounds-checking build traps (NaN positions sample nothing
  // and compare false)
  BEVR_ASSERT(!(live && (r.ys >= (float)(TAP_R - 1) || r.xs >= (float)(TAP_C - 1))));
  rec_out[idx] = r;
  if ((lane & 31) == 0) {
    StepBox sb;
    sb.amin = amin; sb.amax = amax; sb.bmin = bmin; sb.bmax = bmax;
    box_out[2 * gw + (lane >> 5)] = sb;
  }
}

}  // namespace

extern "C" size_t bevr_attn_tap_ws_bytes(const bevr_attn_desc* d) {
  if (bevr_check_desc(d)) return 0;
  return tap_ws_box_offset(*d) + (size_t)d->n_prob * (d->Np / 32) * sizeof(StepBox);
}

extern "C" int bevr_attn_tap_prep(const bevr_attn_desc* d, const float* key_a, const float* key_b, const float* key_y,
                                  const float* key_x, void* tap_ws, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!key_a || !key_b || !key_y || !key_x || !tap_ws) return BEVR_E_NULL;
  if (d->groups != 1) return BEVR_E_SHAPE;
  if (!bevr_aligned16(tap_ws)) return BEVR_E_ALIGN;
  const int n_wave = d->n_prob * (d->Np / KT);
  TapRec* rec = reinterpret_cast<TapRec*>(tap_ws);
  StepBox* box = reinterpret_cast<StepBox*>(reinterpret_cast<char*>(tap_ws) + tap_ws_box_offset(*d));
  hipLaunchKernelGGL(attn_tap_prep_kernel, dim3((n_wave + 3) / 4), dim3(256), 0, (hipStream_t)stream, *d, key_a, key_b,
                     key_y, key_x, rec, box, n_wave);
  return (int)hipGetLastError();
}
