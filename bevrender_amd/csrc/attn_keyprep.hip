// Key preparation for the query-stationary attention kernels (forward, query-side backward).
// One wave per 64-key step: clamps the keys' table coordinates, splits them into the integer row / fractional
// parts the kernels consume (KeyW) and reduces the bounding box of each 32-key half of the step (StepBox; the
// reductions run inside a 32-lane half).  Done once per attention call
// instead of once per (query tile, step) inside the kernels, where the reductions sat on every workgroup's
// per-step critical path.
#include "attn_tile.h"

namespace {

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = min(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = max(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = fminf(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}

__global__ __launch_bounds__(256) void attn_keyprep_kernel(bevr_attn_desc d, const float* __restrict__ key_a,
                                                           const float* __restrict__ key_b, KeyW* __restrict__ kw_out,
                                                           StepBox* __restrict__ box_out, StepBox* __restrict__ gbox_out,
                                                           int n_wave_total) {
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);   // global wave = (prob * groups + grp) * n_step + step
  if (gw >= n_wave_total) return;
  const int lane = threadIdx.x & 63;
  const int n_step = d.Np / KT;
  const int pg = gw / n_step, step = gw % n_step;
  const size_t idx = (size_t)pg * d.Np + (size_t)step * KT + lane;
  const bool live = step * KT + lane < d.N;
  float a = key_a[idx], b = key_b[idx];
  const float aL = -(float)(d.Sp + 1), aU = (float)(d.Ht + 1);
  const float half = (float)(d.Wt / 2);
  const float bL = -(half + 2.0f), bU = (float)(d.Wt + 1);
  a = fminf(fmaxf(a, aL), aU);
  b = fminf(fmaxf(b, bL), bU);
  const float af = floorf(a);
  const int A = (int)af;
  const int amin = wave_min_i(live ? A : 0x7fffffff), amax = wave_max_i(live ? A : (int)0x80000000);
  const float bmin = wave_min_f(live ? b : 3.0e38f), bmax = wave_max_f(live ? b : -3.0e38f);
  KeyW k;
  k.aoff = ((A + d.y_off) + d.x_off * d.Hp) * 8;
  k.fy = a - af;
  // dead (padded) keys: any in-window column (their logits are masked anyway); a half without any live key has no
  // box, so pin them to column 0 rather than to the reduction's identity
  k.b = live ? b : (amax >= amin ? bmin : 0.f);
  // groups of the half (attn_tile.h): row band x column band; only consulted when the half's own box fits no region
  const float gwid = group_width(d);
  const bool groupable = amax >= amin && amax - amin <= 63 && gwid >= 1.0f && (bmax - bmin) < 4.0f * gwid;
  int gid = 0;
  if (live && groupable) {
    const int ga = (A - amin) >> 5;
    const int gb = min(3, max(0, (int)floorf((b - bmin) / gwid)));
    gid = ga * 4 + gb;
  }
  k.arow8 = live ? (A - amin) * 8 + gid : 0;
  kw_out[idx] = k;
  if ((lane & 31) == 0) {
    StepBox sb;
    sb.amin = amin; sb.amax = amax; sb.bmin = bmin; sb.bmax = bmax;
    box_out[2 * gw + (lane >> 5)] = sb;
  }
  StepBox* gb_half = gbox_out + (size_t)(2 * gw + (lane >> 5)) * N_GROUP;
#pragma unroll
  for (int g = 0; g < N_GROUP; ++g) {
    const bool in = live && groupable && gid == g;
    StepBox sb;
    sb.amin = wave_min_i(in ? A : 0x7fffffff);
    sb.amax = wave_max_i(in ? A : (int)0x80000000);
    sb.bmin = wave_min_f(in ? b : 3.0e38f);
    sb.bmax = wave_max_f(in ? b : -3.0e38f);
    if (g == 0 && !groupable) { sb.amin = GROUPS_NONE; sb.amax = (int)0x80000000; }   // no groups: global path
    if ((lane & 31) == 0) gb_half[g] = sb;
  }
}

}  // namespace

extern "C" size_t bevr_attn_key_ws_bytes(const bevr_attn_desc* d) {
  if (bevr_check_desc(d)) return 0;
  return key_ws_gbox_offset(*d) + (size_t)d->n_prob * d->groups * (d->Np / 32) * N_GROUP * sizeof(StepBox);
}

extern "C" int bevr_attn_key_prep(const bevr_attn_desc* d, const float* key_a, const float* key_b, void* key_ws,
                                  void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!key_a || !key_b || !key_ws) return BEVR_E_NULL;
  if (!bevr_aligned16(key_ws)) return BEVR_E_ALIGN;
  const int n_wave = d->n_prob * d->groups * (d->Np / KT);
  KeyW* kw = reinterpret_cast<KeyW*>(key_ws);
  StepBox* box = reinterpret_cast<StepBox*>(reinterpret_cast<char*>(key_ws) + key_ws_box_offset(*d));
  StepBox* gbox = reinterpret_cast<StepBox*>(reinterpret_cast<char*>(key_ws) + key_ws_gbox_offset(*d));
  hipLaunchKernelGGL(attn_keyprep_kernel, dim3((n_wave + 3) / 4), dim3(256), 0, (hipStream_t)stream, *d, key_a, key_b,
                     kw, box, gbox, n_wave);
  return (int)hipGetLastError();
}
