// LayerNorm over the channel axis of channels-last rows (rows x C floats), forward and backward.
//
// Reference: LayerNormProxy (model/model_utils.py:37-49: nn.LayerNorm(C) on the NHWC view) -- the ONE norm an
// EncoderLayer shares between its four uses (model/encoder.py:275), plus the offset heads' norms that are not fused.
// ATen's layer_norm launches one workgroup per row; at C = 64 (256 bytes per row) that ran at 0.55 TB/s forward and
// took three kernels backward.  Here C / 4 lanes own a row (16-byte loads; 4 rows per wave at C = 64), statistics are
// shuffle reductions inside the lane group, and the backward accumulates d(gamma), d(beta) in registers over a
// grid-stride loop, reduces the workgroup's row slots through LDS and sends one atomic per channel and workgroup.
// HBM: forward one read + one write, backward two reads + one write.
#include "bevr_common.h"

namespace {

constexpr int LN_THREADS = 256;

__device__ __forceinline__ float group_sum(float v, int c4n) {
  for (int sh = c4n >> 1; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
  return v;
}

__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ y,
                                                                   float* __restrict__ mean, float* __restrict__ rstd,
                                                                   long long rows, int C, float eps) {
  const int c4n = C >> 2, c4 = threadIdx.x % c4n, slots = LN_THREADS / c4n;
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c4 * 4), b4 = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
  const float inv_c = 1.0f / (float)C;
  // whole lane groups stay converged: the row index is padded, dead groups compute on row 0 and do not store
  const long long n_it = (rows + (long long)gridDim.x * slots - 1) / ((long long)gridDim.x * slots);
  for (long long it = 0; it < n_it; ++it) {
    const long long row = (it * gridDim.x + blockIdx.x) * slots + threadIdx.x / c4n;
    const bool live = row < rows;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (live ? row : 0) * C + c4 * 4);
    const float mu = group_sum(v[0] + v[1] + v[2] + v[3], c4n) * inv_c;
    const f32x4 d = v - f32x4{mu, mu, mu, mu};
    const float var = group_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3], c4n) * inv_c;
    const float rs = rsqrtf(var + eps);
    if (live) {
      *reinterpret_cast<f32x4*>(y + row * C + c4 * 4) = d * rs * g4 + b4;
      if (c4 == 0) {
        mean[row] = mu;
        rstd[row] = rs;
      }
    }
  }
}

__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ dy, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, float* __restrict__ dx,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   long long rows, int C) {
  __shared__ f32x4 red[2][LN_THREADS];
  const int c4n = C >> 2, c4 = threadIdx.x % c4n, slots = LN_THREADS / c4n;
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
  const float inv_c = 1.0f / (float)C;
  f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
  const long long n_it = (rows + (long long)gridDim.x * slots - 1) / ((long long)gridDim.x * slots);
  for (long long it = 0; it < n_it; ++it) {
    const long long row = (it * gridDim.x + blockIdx.x) * slots + threadIdx.x / c4n;
    const bool live = row < rows;
    const long long r0 = live ? row : 0;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r0 * C + c4 * 4);
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + r0 * C + c4 * 4);
    if (!live) g = f32x4{0.f, 0.f, 0.f, 0.f};
    const float mu = mean[r0], rs = rstd[r0];
    const f32x4 xh = (v - f32x4{mu, mu, mu, mu}) * rs;
    const f32x4 gg = g * g4;
    const float m1 = group_sum(gg[0] + gg[1] + gg[2] + gg[3], c4n) * inv_c;
    const float m2 = group_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3], c4n) * inv_c;
    if (live) *reinterpret_cast<f32x4*>(dx + row * C + c4 * 4) = (gg - f32x4{m1, m1, m1, m1} - xh * m2) * rs;
    ag += g * xh;
    ab += g;
  }
  red[0][threadIdx.x] = ag;
  red[1][threadIdx.x] = ab;
  __syncthreads();
  for (int u = threadIdx.x; u < 2 * c4n; u += LN_THREADS) {
    const int which = u / c4n, q = u % c4n;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < slots; ++sl) s += red[which][sl * c4n + q];
    float* dst = (which ? dbeta : dgamma) + q * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(dst + k, s[k]);
  }
}

int check(long long rows, int C) {
  const int c4n = C >> 2;
  if (rows <= 0 || C <= 0 || (C & 3) || c4n > 64 || (c4n & (c4n - 1))) return BEVR_E_SHAPE;
  return BEVR_OK;
}

int grid_of(long long rows, int C) {
  const long long per = LN_THREADS / (C >> 2);
  long long g = (rows + per - 1) / per;
  if (g > 256 * 8) g = 256 * 8;   // grid-stride the rest: the backward's per-workgroup reduction stays amortised
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int bevr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, long long rows, int C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return BEVR_E_NULL;
  int rc = check(rows, C);
  if (rc) return rc;
  if (!bevr_aligned16(x) || !bevr_aligned16(y) || !bevr_aligned16(gamma) || !bevr_aligned16(beta)) return BEVR_E_ALIGN;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid_of(rows, C)), dim3(LN_THREADS), 0, (hipStream_t)stream, x, gamma, beta, y,
                     mean, rstd, rows, C, eps);
  return (int)hipGetLastError();
}

// dgamma, dbeta [C] are ACCUMULATED (the caller zeroes them); dx is written
extern "C" int bevr_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* mean, const float* rstd,
                                  float* dx, float* dgamma, float* dbeta, long long rows, int C, void* stream) {
  if (!x || !gamma || !dy || !mean || !rstd || !dx || !dgamma || !dbeta) return BEVR_E_NULL;
  int rc = check(rows, C);
  if (rc) return rc;
  if (!bevr_aligned16(x) || !bevr_aligned16(dy) || !bevr_aligned16(dx) || !bevr_aligned16(gamma)) return BEVR_E_ALIGN;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid_of(rows, C)), dim3(LN_THREADS), 0, (hipStream_t)stream, x, gamma, dy, mean,
                     rstd, dx, dgamma, dbeta, rows, C);
  return (int)hipGetLastError();
}
