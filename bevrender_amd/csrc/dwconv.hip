// Depthwise k x k convolution (one filter per channel, stride 1, "same" zero padding, odd k) -- the local
// perception units and the MLP's 3x3 of the reference's EncoderLayer (model/encoder.py:363-411,
// model/model_utils.py:6-35) and TSA's offset head (model/TSA_deform_attn.py:54-68).  MIOpen runs the fp32 weight
// gradient of these through a naive reference kernel (59 ms per call at 200x200, C = 256) and the shifted
// multiply-add formulation in PyTorch needs ~45 bandwidth-bound passes; here forward and input gradient are one
// read + one write, the weight gradient one read of each operand.
//
// One layout-generic set of kernels: the tensor is addressed as x[b*sb + y*sy + xx*sx + c*sc] with the UNIT-stride
// axis on the lanes (NHWC: channels, NCHW: the image row), so every load is coalesced.  HBM bound:
// (1 + 1) * numel * 4 bytes per pass; the k*k taps come from L1/L2.
#include "bevr_common.h"

namespace {

struct DwGeom {
  int B, H, W, C, k;
  long sb, sy, sx, sc;   // element strides
  int nhwc;
};

// forward / input gradient (flip = 1: correlate with the flipped filter)
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwGeom g, const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int flip) {
  const long n = (long)g.B * g.H * g.W * g.C;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  int c, xx, yy, b;
  if (g.nhwc) { c = idx % g.C; long t = idx / g.C; xx = t % g.W; t /= g.W; yy = t % g.H; b = t / g.H; }
  else { xx = idx % g.W; long t = idx / g.W; yy = t % g.H; t /= g.H; c = t % g.C; b = t / g.C; }
  const int p = g.k / 2;
  const float* wc = w + (long)c * g.k * g.k;
  const float* xb = x + b * g.sb + c * g.sc;
  float acc = bias ? bias[c] : 0.f;
  for (int dy = 0; dy < g.k; ++dy) {
    const int iy = yy + dy - p;
    if (iy < 0 || iy >= g.H) continue;
    for (int dx = 0; dx < g.k; ++dx) {
      const int ix = xx + dx - p;
      if (ix < 0 || ix >= g.W) continue;
      const float wv = flip ? wc[(g.k - 1 - dy) * g.k + (g.k - 1 - dx)] : wc[dy * g.k + dx];
      acc = fmaf(xb[iy * g.sy + ix * g.sx], wv, acc);
    }
  }
  y[b * g.sb + yy * g.sy + xx * g.sx + c * g.sc] = acc;
}

// weight and bias gradient: dw[c][dy][dx] += sum_{b,y,x} dy_[b,y,x,c] * x[b,y+dy-p,x+dx-p,c]
// A thread owns one line of the tensor along the NON-unit spatial axis (NHWC: a (b, y, c) line over x;
// NCHW: a (b, c, x) line over y) and keeps the k*k partial sums in registers; NCHW lanes share the channel, so the
// wave reduces before the atomics.  KMAX: k <= 5.
constexpr int KK_MAX = 25;
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(DwGeom g, const float* __restrict__ x, const float* __restrict__ dy_,
                                                           float* __restrict__ dw, float* __restrict__ dbias) {
  const int p = g.k / 2, kk = g.k * g.k;
  const long n_line = g.nhwc ? (long)g.B * g.H * g.C : (long)g.B * g.C * g.W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const bool on = idx < n_line;
  int c = 0, b = 0, fix = 0;
  if (on) {
    if (g.nhwc) { c = idx % g.C; long t = idx / g.C; fix = t % g.H; b = t / g.H; }      // fix = y, run over x
    else { fix = idx % g.W; long t = idx / g.W; c = t % g.C; b = t / g.C; }             // fix = x, run over y
  }
  float acc[KK_MAX];
#pragma unroll
  for (int t = 0; t < KK_MAX; ++t) acc[t] = 0.f;
  float accb = 0.f;
  if (on) {
    const float* xb = x + b * g.sb + c * g.sc;
    const float* db = dy_ + b * g.sb + c * g.sc;
    const int n_run = g.nhwc ? g.W : g.H;
    for (int r = 0; r < n_run; ++r) {
      const int yy = g.nhwc ? fix : r, xx = g.nhwc ? r : fix;
      const float gq = db[yy * g.sy + xx * g.sx];
      accb += gq;
#pragma unroll
      for (int t = 0; t < KK_MAX; ++t) {
        if (t < kk) {
          const int iy = yy + t / g.k - p, ix = xx + t % g.k - p;
          if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) acc[t] = fmaf(gq, xb[iy * g.sy + ix * g.sx], acc[t]);
        }
      }
    }
  }
  if (g.nhwc) {
    if (on) {
      for (int t = 0; t < kk; ++t) atomicAdd(dw + (long)c * kk + t, acc[t]);
      if (dbias) atomicAdd(dbias + c, accb);
    }
  } else {
    // lanes of a wave: consecutive x of (mostly) one channel -> reduce lanes with equal channel via shuffles when
    // the whole wave shares it, else fall back to per-lane atomics
    const int c0 = __shfl(c, 0);
    const bool same = __all(!on || c == c0);
    if (same) {
#pragma unroll
      for (int t = 0; t < KK_MAX; ++t) {
        if (t < kk) {
          float v = on ? acc[t] : 0.f;
          for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
          if ((threadIdx.x & 63) == 0) atomicAdd(dw + (long)c0 * kk + t, v);
        }
      }
      float v = on ? accb : 0.f;
      for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
      if (dbias && (threadIdx.x & 63) == 0) atomicAdd(dbias + c0, v);
    } else if (on) {
      for (int t = 0; t < kk; ++t) atomicAdd(dw + (long)c * kk + t, acc[t]);
      if (dbias) atomicAdd(dbias + c, accb);
    }
  }
}

DwGeom make_geom(int B, int H, int W, int C, int k, int nhwc) {
  DwGeom g;
  g.B = B; g.H = H; g.W = W; g.C = C; g.k = k; g.nhwc = nhwc;
  if (nhwc) { g.sc = 1; g.sx = C; g.sy = (long)W * C; g.sb = (long)H * W * C; }
  else { g.sx = 1; g.sy = W; g.sc = (long)H * W; g.sb = (long)C * H * W; }
  return g;
}

}  // namespace

extern "C" int bevr_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                               int k, int nhwc, int flip, void* stream) {
  if (!x || !w || !y) return BEVR_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || k > 5 || (k & 1) == 0) return BEVR_E_SHAPE;
  const DwGeom g = make_geom(B, H, W, C, k, nhwc);
  const long n = (long)B * H * W * C;
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, x, w, bias, y,
                     flip);
  return (int)hipGetLastError();
}

extern "C" int bevr_dwconv_bwd_w(const float* x, const float* dy, float* dw, float* dbias, int B, int H, int W, int C, int k,
                                 int nhwc, void* stream) {
  if (!x || !dy || !dw) return BEVR_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || k > 5 || (k & 1) == 0) return BEVR_E_SHAPE;
  const DwGeom g = make_geom(B, H, W, C, k, nhwc);
  const long n_line = nhwc ? (long)B * H * C : (long)B * C * W;
  hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3((unsigned)((n_line + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, x, dy,
                     dw, dbias);
  return (int)hipGetLastError();
}
