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

// ---- channels-last fast paths (C % 4 == 0): the MLP's 3x3 on the 4x-expanded BEV (C = 256, 328 MB per tensor at the
// benchmark size) is two thirds of the depthwise time.  The generic kernel above reads every input element k*k times
// through L1 / L2 and pays four 64-bit divisions per element; here a thread owns 4 channels x DW_XT consecutive pixels of
// one row: (DW_XT + K - 1) 16-byte loads per filter row feed DW_XT outputs (3.75 loads per output at K = 3 against 9),
// filter taps in registers.
constexpr int DW_XT = 8;

// MODE (the MLP's  act(y + dwc(y))  of model/model_utils.py:51-59 as one kernel each way, bevr_dwconv_res_gelu):
//   0  y = conv(x) + bias                                   1  y = gelu(x + conv(x) + bias)
//   2  y = aux * gelu'(x + conv(x) + bias)   (backward: the gradient at the pre-activation, recomputed, aux = d out)
//   3  y = x + conv(x)                       (backward, flip = 1: the input gradient g + conv^T(g))
template <int K, int MODE>
__global__ __launch_bounds__(256) void dwconv_fwd_nhwc4_kernel(int B, int H, int W, int C, const float* __restrict__ x,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ aux, float* __restrict__ y, int flip) {
  constexpr int P = K / 2;
  const int c4n = C >> 2, n_xt = (W + DW_XT - 1) / DW_XT;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * H * n_xt * c4n) return;
  const int c4 = (int)(idx % c4n);
  long t = idx / c4n;
  const int xt = (int)(t % n_xt);
  t /= n_xt;
  const int yy = (int)(t % H), b = (int)(t / H);
  const int x0 = xt * DW_XT, c0 = c4 * 4;
  f32x4 wv[K * K];   // wv[tap][channel of the quad]
#pragma unroll
  for (int tp = 0; tp < K * K; ++tp) {
    const int src = flip ? K * K - 1 - tp : tp;
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[tp][k] = w[(long)(c0 + k) * K * K + src];
  }
  f32x4 acc[DW_XT], xc[DW_XT];      // xc: the centre input of each output (MODE > 0)
  const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < DW_XT; ++i) acc[i] = b4;
  const float* xb = x + ((long)b * H * W) * C + c0;
#pragma unroll
  for (int dy = 0; dy < K; ++dy) {
    const int iy = yy + dy - P;
    if (iy < 0 || iy >= H) continue;
    f32x4 row[DW_XT + K - 1];
#pragma unroll
    for (int i = 0; i < DW_XT + K - 1; ++i) {
      const int ix = x0 + i - P;
      row[i] = (ix >= 0 && ix < W) ? *reinterpret_cast<const f32x4*>(xb + ((long)iy * W + ix) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < DW_XT; ++i)
#pragma unroll
      for (int dx = 0; dx < K; ++dx) acc[i] += row[i + dx] * wv[dy * K + dx];
    if (MODE > 0 && dy == P) {
#pragma unroll
      for (int i = 0; i < DW_XT; ++i) xc[i] = row[i + P];
    }
  }
  float* yb = y + (((long)b * H + yy) * W) * C + c0;
  const float* ab = MODE == 2 ? aux + (((long)b * H + yy) * W) * C + c0 : nullptr;
#pragma unroll
  for (int i = 0; i < DW_XT; ++i) {
    if (x0 + i >= W) continue;
    f32x4 o = acc[i];
    if constexpr (MODE > 0) {
      o += xc[i];
      if constexpr (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = gelu_erf(o[k]);
      } else if constexpr (MODE == 2) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ab + (long)(x0 + i) * C);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = a[k] * gelu_erf_grad(o[k]);
      }
    }
    *reinterpret_cast<f32x4*>(yb + (long)(x0 + i) * C) = o;
  }
}

// weight / bias gradient, channels-last.  A wave takes one image row: lane = (channel quad, x segment) -- 64 quads at
// C = 256, 16 quads x 4 segments at C = 64 -- and slides a K x K window of 16-byte loads along its segment with the
// K*K*4 partial sums in registers; the 4 waves of a workgroup (4 rows) and the x segments are summed through LDS, and
// one atomic per (channel, tap) leaves the WORKGROUP: 4-16x fewer atomics onto the same C*K*K addresses than one per
// row line.
template <int K>
__global__ __launch_bounds__(256) void dwconv_bwd_w_nhwc4_kernel(int B, int H, int W, int C, const float* __restrict__ x,
                                                                 const float* __restrict__ dy_, float* __restrict__ dw,
                                                                 float* __restrict__ dbias) {
  constexpr int P = K / 2, NV = K * K + 1;            // partial sums per lane: taps + bias, each a channel quad
  __shared__ f32x4 red[4][NV][64];
  const int c4n = C >> 2;
  const int qpw = min(c4n, 64), n_seg = 64 / qpw;     // quads per wave, x segments per wave (c4n is a power of two or >= 64)
  const int n_qg = (c4n + qpw - 1) / qpw;             // quad groups (C > 256: several workgroups per row block)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qg = blockIdx.x % n_qg;
  const long rb = blockIdx.x / n_qg;                  // row block: 4 consecutive rows of one image
  const int n_rb = (H + 3) / 4;
  const int b = (int)(rb / n_rb), yy = (int)(rb % n_rb) * 4 + wave;
  const int quad = qg * qpw + lane % qpw, seg = lane / qpw;
  const bool on = yy < H && quad < c4n;
  const int c0 = quad * 4;
  const int seg_w = (W + n_seg - 1) / n_seg, xa = seg * seg_w, xe = min(W, xa + seg_w);
  f32x4 acc[NV];
#pragma unroll
  for (int tp = 0; tp < NV; ++tp) acc[tp] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (on) {
    const float* xb = x + ((long)b * H * W) * C + c0;
    const float* gb = dy_ + (((long)b * H + yy) * W) * C + c0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 win[K][K];                                  // win[dy][j] = x[yy + dy - P][xx + j - P]
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int iy = yy + dy - P, ix = xa + j - P;
        win[dy][j] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? *reinterpret_cast<const f32x4*>(xb + ((long)iy * W + ix) * C) : z;
      }
    for (int xx = xa; xx < xe; ++xx) {
      const f32x4 gq = *reinterpret_cast<const f32x4*>(gb + (long)xx * C);
      acc[K * K] += gq;
#pragma unroll
      for (int dy = 0; dy < K; ++dy)
#pragma unroll
        for (int j = 0; j < K; ++j) acc[dy * K + j] += gq * win[dy][j];
      const int nx = xx + 1 + P;                      // shift the window one pixel to the right
#pragma unroll
      for (int dy = 0; dy < K; ++dy) {
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) win[dy][j] = win[dy][j + 1];
        const int iy = yy + dy - P;
        win[dy][K - 1] = (iy >= 0 && iy < H && nx < W) ? *reinterpret_cast<const f32x4*>(xb + ((long)iy * W + nx) * C) : z;
      }
    }
  }
#pragma unroll
  for (int tp = 0; tp < NV; ++tp) red[wave][tp][lane] = acc[tp];
  __syncthreads();
  // (quad of this workgroup, value): sum over the 4 waves and the x segments, one atomic each
  for (int u = threadIdx.x; u < qpw * NV; u += 256) {
    const int q = u % qpw, tp = u / qpw;
    if (qg * qpw + q >= c4n) continue;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int wv = 0; wv < 4; ++wv)
      for (int sg = 0; sg < n_seg; ++sg) sum += red[wv][tp][sg * qpw + q];
    const int cc = (qg * qpw + q) * 4;
    if (tp < K * K) {
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(dw + (long)(cc + k) * K * K + tp, sum[k]);
    } else if (dbias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(dbias + cc + k, sum[k]);
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
  if (nhwc && (C & 3) == 0 && (k == 3 || k == 5) && bevr_aligned16(x) && bevr_aligned16(y) && (!bias || bevr_aligned16(bias))) {
    const long nthr = (long)B * H * ((W + DW_XT - 1) / DW_XT) * (C >> 2);
    const dim3 grid((unsigned)((nthr + 255) / 256));
    if (k == 3)
      hipLaunchKernelGGL((dwconv_fwd_nhwc4_kernel<3, 0>), grid, dim3(256), 0, (hipStream_t)stream, B, H, W, C, x, w, bias,
                         (const float*)nullptr, y, flip);
    else
      hipLaunchKernelGGL((dwconv_fwd_nhwc4_kernel<5, 0>), grid, dim3(256), 0, (hipStream_t)stream, B, H, W, C, x, w, bias,
                         (const float*)nullptr, y, flip);
    return (int)hipGetLastError();
  }
  const DwGeom g = make_geom(B, H, W, C, k, nhwc);
  const long n = (long)B * H * W * C;
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, x, w, bias, y,
                     flip);
  return (int)hipGetLastError();
}

// The MLP's  act(y + dwc(y))  (model/model_utils.py:51-59, GELU in its erf form) fused with the depthwise 3 x 3,
// channels-last: mode 1 the forward, modes 2 and 3 the two backward kernels (see dwconv_fwd_nhwc4_kernel).
extern "C" int bevr_dwconv_res_gelu(const float* x, const float* w, const float* bias, const float* aux, float* y, int B, int H,
                                    int W, int C, int k, int mode, void* stream) {
  if (!x || !w || !y || (mode == 2 && !aux)) return BEVR_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || k != 3 || mode < 1 || mode > 3) return BEVR_E_SHAPE;
  if (!bevr_aligned16(x) || !bevr_aligned16(y) || (bias && !bevr_aligned16(bias)) || (aux && !bevr_aligned16(aux)))
    return BEVR_E_ALIGN;
  const long nthr = (long)B * H * ((W + DW_XT - 1) / DW_XT) * (C >> 2);
  const dim3 grid((unsigned)((nthr + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (mode == 1)
    hipLaunchKernelGGL((dwconv_fwd_nhwc4_kernel<3, 1>), grid, dim3(256), 0, st, B, H, W, C, x, w, bias, aux, y, 0);
  else if (mode == 2)
    hipLaunchKernelGGL((dwconv_fwd_nhwc4_kernel<3, 2>), grid, dim3(256), 0, st, B, H, W, C, x, w, bias, aux, y, 0);
  else
    hipLaunchKernelGGL((dwconv_fwd_nhwc4_kernel<3, 3>), grid, dim3(256), 0, st, B, H, W, C, x, w, (const float*)nullptr, aux, y, 1);
  return (int)hipGetLastError();
}

// NCHW, 3 x 3, W <= 256: a workgroup owns a band of rows of one (b, c) plane, thread = column x; the three input rows a
// tap reaches slide through registers, so a row of x is read once per band (three coalesced loads per thread: x - 1, x,
// x + 1) instead of once per tap, and the workgroup's ten sums leave through LDS as ten atomics (the line kernel above
// reads every input 9 times and, where W is not a multiple of 64, falls back to per-lane atomics: 0.72 ms per call at
// 8 x 64 x 200 x 200).
constexpr int DW_BANDS = 4;
__global__ __launch_bounds__(256) void dwconv_bwd_w_nchw3_kernel(int H, int W, const float* __restrict__ x,
                                                                const float* __restrict__ dy_, float* __restrict__ dw,
                                                                float* __restrict__ dbias, int C) {
  __shared__ float red[4][10];
  const int plane = blockIdx.x, band = blockIdx.y, xx = threadIdx.x;
  const int c = plane % C;
  const int rows = (H + DW_BANDS - 1) / DW_BANDS, y0 = band * rows, y1 = min(H, y0 + rows);
  const float* xp = x + (size_t)plane * H * W;
  const float* dp = dy_ + (size_t)plane * H * W;
  const bool on = xx < W;
  auto row3 = [&](int yy, float (&r)[3]) {      // x[yy][xx - 1 .. xx + 1], zero outside the plane
    const bool ry = on && yy >= 0 && yy < H;
    r[0] = (ry && xx >= 1) ? xp[(size_t)yy * W + xx - 1] : 0.f;
    r[1] = ry ? xp[(size_t)yy * W + xx] : 0.f;
    r[2] = (ry && xx + 1 < W) ? xp[(size_t)yy * W + xx + 1] : 0.f;
  };
  float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, accb = 0.f;
  float ra[3], rb[3], rc[3];
  row3(y0 - 1, ra);
  row3(y0, rb);
  for (int yy = y0; yy < y1; ++yy) {
    row3(yy + 1, rc);
    const float gq = on ? dp[(size_t)yy * W + xx] : 0.f;
    accb += gq;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      acc[t] = fmaf(gq, ra[t], acc[t]);
      acc[3 + t] = fmaf(gq, rb[t], acc[3 + t]);
      acc[6 + t] = fmaf(gq, rc[t], acc[6 + t]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) { ra[t] = rb[t]; rb[t] = rc[t]; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    float v = t < 9 ? acc[t] : accb;
    for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
    if (lane == 0) red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (threadIdx.x < 9) atomicAdd(dw + (size_t)c * 9 + threadIdx.x, v);
    else if (dbias) atomicAdd(dbias + c, v);
  }
}

extern "C" int bevr_dwconv_bwd_w(const float* x, const float* dy, float* dw, float* dbias, int B, int H, int W, int C, int k,
                                 int nhwc, void* stream) {
  if (!x || !dy || !dw) return BEVR_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || k > 5 || (k & 1) == 0) return BEVR_E_SHAPE;
  const int c4n = C >> 2;
  if (nhwc && (C & 3) == 0 && k == 3 && (c4n >= 64 || (c4n & (c4n - 1)) == 0) && bevr_aligned16(x) && bevr_aligned16(dy)) {
    const int qpw = c4n < 64 ? c4n : 64;
    const long nblk = (long)B * ((H + 3) / 4) * ((c4n + qpw - 1) / qpw);
    hipLaunchKernelGGL(dwconv_bwd_w_nhwc4_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, B, H, W, C, x, dy,
                       dw, dbias);
    return (int)hipGetLastError();
  }
  if (!nhwc && k == 3 && W <= 256) {
    hipLaunchKernelGGL(dwconv_bwd_w_nchw3_kernel, dim3((unsigned)(B * C), DW_BANDS), dim3(256), 0, (hipStream_t)stream, H, W, x,
                       dy, dw, dbias, C);
    return (int)hipGetLastError();
  }
  const DwGeom g = make_geom(B, H, W, C, k, nhwc);
  const long n_line = nhwc ? (long)B * H * C : (long)B * C * W;
  hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3((unsigned)((n_line + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, x, dy,
                     dw, dbias);
  return (int)hipGetLastError();
}
