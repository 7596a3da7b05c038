// Offset heads of the deformable attention blocks, fused: per BEV pixel
//   z[c*Mx + m] = x[c] * w0[c*Mx + m] + b0[c*Mx + m]        (depthwise 1x1 with channel multiplier Mx; optional)
//   y = LayerNorm_{C*Mx}(z) * gamma + beta ;  a = GELU(y) (erf form) ;  out[d] = sum_k W3[d][k] a[k]   (1x1, no bias)
// Replaces model/SCA_deform_attn.py:56-77 (conv_offset_m{v}: Mx = D, Dout = D) and the LayerNorm -> GELU -> 1x1 tail
// of model/TSA_deform_attn.py:54-68 (Mx = 1 without w0/b0, Dout = 2), forward and backward.
//
// The stock-op chain materialises the (B, S, S, C*D) expansion three times per view (expanded, normalised,
// activated): 320 floats per pixel and view at C = 64, D = 5, against 64 read and 5 written here.  HBM-bound by
// construction: one wave per pixel, lane = input channel, the C*Mx expanded values of a pixel never leave the
// wave's registers; LayerNorm statistics and the Dout dot products are wave reductions.  A workgroup keeps the
// head's parameters in registers for its whole life; the backward accumulates the parameter gradients in registers
// over the workgroup's pixels and adds them once at the end, the input gradient with float atomics (the V views of
// one sample share the input).
#include "bevr_common.h"

namespace {

constexpr int OH_MAXM = 8;   // channel multiplier and output count limits (register arrays)
constexpr int OH_MAXD = 8;
constexpr int OH_THREADS = 256;

// Sum over the 64 lanes, the same value returned to every lane.  Data-parallel-primitive adds instead of six
// ds_bpermute round trips (7 reductions per pixel in the forward: 42 LDS crossbar instructions): quad swaps, the two row
// mirrors (every lane of a 16-lane row then holds the row's sum), row_bcast:15 / :31 to chain the four rows (masked rows add
// the `old` operand, 0), and the total read off lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1, 0, 3, 2]
  v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2, 3, 0, 1]
  v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);     // row_mirror
  v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct HeadParams {   // this lane's slice of the head: channel c = lane, expanded channels c*Mx .. c*Mx + Mx - 1
  float w0[OH_MAXM], b0[OH_MAXM], ga[OH_MAXM], be[OH_MAXM], w3[OH_MAXD][OH_MAXM];
};

template <int Mx, int Dout>
__device__ __forceinline__ void load_params(HeadParams& p, const float* w0, const float* b0, const float* gamma,
                                            const float* beta, const float* W3, int c, int Cg, bool on) {
  const int K = Cg * Mx;
#pragma unroll
  for (int m = 0; m < Mx; ++m) {
    const int k = c * Mx + m;
    p.w0[m] = on ? (w0 ? w0[k] : 1.0f) : 0.f;
    p.b0[m] = on && b0 ? b0[k] : 0.f;
    p.ga[m] = on ? gamma[k] : 0.f;
    p.be[m] = on ? beta[k] : 0.f;
#pragma unroll
    for (int d = 0; d < Dout; ++d) p.w3[d][m] = on ? W3[(size_t)d * K + k] : 0.f;
  }
}

// forward quantities of one pixel for this lane
template <int Mx>
struct PixFwd { float xh[Mx], y[Mx], a[Mx]; float rstd; };

template <int Mx>
__device__ __forceinline__ void pixel_forward(PixFwd<Mx>& f, const HeadParams& p, float xc, bool on, float inv_k, float eps) {
  float z[Mx], s = 0.f;
#pragma unroll
  for (int m = 0; m < Mx; ++m) { z[m] = on ? fmaf(xc, p.w0[m], p.b0[m]) : 0.f; s += z[m]; }
  const float mean = wave_sum(s) * inv_k;
  float q = 0.f;
#pragma unroll
  for (int m = 0; m < Mx; ++m) { const float dlt = on ? z[m] - mean : 0.f; q += dlt * dlt; }
  f.rstd = rsqrtf(wave_sum(q) * inv_k + eps);
#pragma unroll
  for (int m = 0; m < Mx; ++m) {
    f.xh[m] = on ? (z[m] - mean) * f.rstd : 0.f;
    f.y[m] = fmaf(f.xh[m], p.ga[m], p.be[m]);
    f.a[m] = on ? gelu_erf(f.y[m]) : 0.f;
  }
}

template <int Mx, int Dout>
__global__ __launch_bounds__(OH_THREADS) void offset_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                                     const float* __restrict__ b0, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ W3,
                                                                     float* __restrict__ out, long long P, int Cg, int xstride,
                                                                     float eps) {
  const int lane = threadIdx.x & 63;
  const bool on = lane < Cg;
  HeadParams p;
  load_params<Mx, Dout>(p, w0, b0, gamma, beta, W3, lane, Cg, on);
  const float inv_k = 1.0f / (float)(Cg * Mx);
  const long long wave0 = (long long)blockIdx.x * (OH_THREADS / 64) + (threadIdx.x >> 6);
  const long long nwave = (long long)gridDim.x * (OH_THREADS / 64);
  for (long long pix = wave0; pix < P; pix += nwave) {
    const float xc = on ? x[pix * xstride + lane] : 0.f;
    PixFwd<Mx> f;
    pixel_forward<Mx>(f, p, xc, on, inv_k, eps);
    float o[Dout];
#pragma unroll
    for (int d = 0; d < Dout; ++d) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < Mx; ++m) s = fmaf(p.w3[d][m], f.a[m], s);
      o[d] = wave_sum(s);
    }
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < Dout; ++d) out[pix * Dout + d] = o[d];
    }
  }
}

template <int Mx, int Dout>
__global__ __launch_bounds__(OH_THREADS) void offset_head_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ W3, const float* __restrict__ dout, float* __restrict__ dx,
    float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dW3, long long P, int Cg, int xstride, float eps) {
  const int lane = threadIdx.x & 63;
  const bool on = lane < Cg;
  HeadParams p;
  load_params<Mx, Dout>(p, w0, b0, gamma, beta, W3, lane, Cg, on);
  const float inv_k = 1.0f / (float)(Cg * Mx);
  float gw0[Mx], gb0[Mx], gga[Mx], gbe[Mx], gw3[Dout][Mx];
#pragma unroll
  for (int m = 0; m < Mx; ++m) {
    gw0[m] = gb0[m] = gga[m] = gbe[m] = 0.f;
#pragma unroll
    for (int d = 0; d < Dout; ++d) gw3[d][m] = 0.f;
  }
  const long long wave0 = (long long)blockIdx.x * (OH_THREADS / 64) + (threadIdx.x >> 6);
  const long long nwave = (long long)gridDim.x * (OH_THREADS / 64);
  for (long long pix = wave0; pix < P; pix += nwave) {
    const float xc = on ? x[pix * xstride + lane] : 0.f;
    PixFwd<Mx> f;
    pixel_forward<Mx>(f, p, xc, on, inv_k, eps);
    float go[Dout];
#pragma unroll
    for (int d = 0; d < Dout; ++d) go[d] = dout[pix * Dout + d];   // same address for every lane: broadcast load
    float dxh[Mx], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < Mx; ++m) {
      float da = 0.f;
#pragma unroll
      for (int d = 0; d < Dout; ++d) { da = fmaf(go[d], p.w3[d][m], da); gw3[d][m] = fmaf(go[d], f.a[m], gw3[d][m]); }
      const float dy = on ? da * gelu_erf_grad(f.y[m]) : 0.f;
      gga[m] = fmaf(dy, f.xh[m], gga[m]);
      gbe[m] += dy;
      dxh[m] = dy * p.ga[m];
      s1 += dxh[m];
      s2 = fmaf(dxh[m], f.xh[m], s2);
    }
    const float m1 = wave_sum(s1) * inv_k, m2 = wave_sum(s2) * inv_k;
    float dxc = 0.f;
#pragma unroll
    for (int m = 0; m < Mx; ++m) {
      const float dz = on ? f.rstd * (dxh[m] - m1 - f.xh[m] * m2) : 0.f;
      gw0[m] = fmaf(dz, xc, gw0[m]);
      gb0[m] += dz;
      dxc = fmaf(dz, p.w0[m], dxc);
    }
    if (on && dx) atomicAdd(dx + pix * xstride + lane, dxc);
  }
  // parameter gradients: summed over the workgroup's waves in LDS, then ONE global atomic per parameter and workgroup
  // (one per wave put 8 192 waves on the same 2 880 addresses: 1.4 ms per launch of same-address contention)
  __shared__ float pg[(4 + OH_MAXD) * 64 * OH_MAXM];
  const int K = Cg * Mx;
  for (int i = threadIdx.x; i < (4 + Dout) * K; i += OH_THREADS) pg[i] = 0.f;
  __syncthreads();
  if (on) {
#pragma unroll
    for (int m = 0; m < Mx; ++m) {
      const int k = lane * Mx + m;
      atomicAdd(pg + k, gw0[m]);
      atomicAdd(pg + K + k, gb0[m]);
      atomicAdd(pg + 2 * K + k, gga[m]);
      atomicAdd(pg + 3 * K + k, gbe[m]);
#pragma unroll
      for (int d = 0; d < Dout; ++d) atomicAdd(pg + (4 + d) * K + k, gw3[d][m]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (4 + Dout) * K; i += OH_THREADS) {
    const int which = i / K, k = i - which * K;
    float* dst = which == 0 ? dw0 : which == 1 ? db0 : which == 2 ? dgamma : which == 3 ? dbeta : dW3 + (size_t)(which - 4) * K;
    if (dst) atomicAdd(dst + k, pg[i]);
  }
}

int oh_grid(long long P) {
  long long g = (P + (OH_THREADS / 64) * 16 - 1) / ((OH_THREADS / 64) * 16);   // >= 16 pixels per wave: amortises the parameter load
  if (g > 256 * 4) g = 256 * 4;   // 4 workgroups per CU: enough waves to hide the loads, few enough parameter-gradient atomics
  if (g < 1) g = 1;
  return (int)g;
}

int oh_check(const float* x, const float* gamma, const float* beta, const float* W3, long long P, int Cg, int xstride, int Mx,
             int Dout) {
  if (!x || !gamma || !beta || !W3) return BEVR_E_NULL;
  if (P <= 0 || Cg <= 0 || Cg > 64 || xstride < Cg || Mx < 1 || Mx > OH_MAXM || Dout < 1 || Dout > OH_MAXD) return BEVR_E_SHAPE;
  return BEVR_OK;
}

// dispatch on (Mx, Dout): the shapes the reference can produce at Cg <= 64 -- SCA (D, D) with D <= 8, TSA (1, 2)
#define OH_DISPATCH(CALL)                                                         \
  if (Mx == 1 && Dout == 2) { CALL(1, 2); }                                        \
  else if (Mx == Dout && Mx == 1) { CALL(1, 1); } else if (Mx == Dout && Mx == 2) { CALL(2, 2); } \
  else if (Mx == Dout && Mx == 3) { CALL(3, 3); } else if (Mx == Dout && Mx == 4) { CALL(4, 4); } \
  else if (Mx == Dout && Mx == 5) { CALL(5, 5); } else if (Mx == Dout && Mx == 6) { CALL(6, 6); } \
  else if (Mx == Dout && Mx == 7) { CALL(7, 7); } else if (Mx == Dout && Mx == 8) { CALL(8, 8); } \
  else return BEVR_E_SHAPE;

}  // namespace

extern "C" int bevr_offset_head_fwd(const float* x, const float* w0, const float* b0, const float* gamma, const float* beta,
                                    const float* W3, float* out, long long P, int Cg, int xstride, int Mx, int Dout, float eps,
                                    void* stream) {
  int rc = oh_check(x, gamma, beta, W3, P, Cg, xstride, Mx, Dout);
  if (rc) return rc;
  if (!out) return BEVR_E_NULL;
#define OH_FWD(M, D)                                                                                                   \
  hipLaunchKernelGGL((offset_head_fwd_kernel<M, D>), dim3(oh_grid(P)), dim3(OH_THREADS), 0, (hipStream_t)stream, x, w0, b0, \
                     gamma, beta, W3, out, P, Cg, xstride, eps)
  OH_DISPATCH(OH_FWD)
#undef OH_FWD
  return (int)hipGetLastError();
}

extern "C" int bevr_offset_head_bwd(const float* x, const float* w0, const float* b0, const float* gamma, const float* beta,
                                    const float* W3, const float* dout, float* dx, float* dw0, float* db0, float* dgamma,
                                    float* dbeta, float* dW3, long long P, int Cg, int xstride, int Mx, int Dout, float eps,
                                    void* stream) {
  int rc = oh_check(x, gamma, beta, W3, P, Cg, xstride, Mx, Dout);
  if (rc) return rc;
  if (!dout || !dgamma || !dbeta || !dW3) return BEVR_E_NULL;
#define OH_BWD(M, D)                                                                                                   \
  hipLaunchKernelGGL((offset_head_bwd_kernel<M, D>), dim3(oh_grid(P)), dim3(OH_THREADS), 0, (hipStream_t)stream, x, w0, b0, \
                     gamma, beta, W3, dout, dx, dw0, db0, dgamma, dbeta, dW3, P, Cg, xstride, eps)
  OH_DISPATCH(OH_BWD)
#undef OH_BWD
  return (int)hipGetLastError();
}
