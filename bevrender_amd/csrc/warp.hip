// Ego-motion warp of the history BEV (reference model/encoder.py:413-466): one torchvision-style affine resampling
//   out = bilinear(img, M (x, y)) * bilinear(ones, M (x, y))          (zero padding, fill = 0)
// for a whole batch in one launch.  The reference calls torchvision.transforms.functional.affine(..., BILINEAR,
// fill=0) per sample in a Python loop with .item() syncs, twice in a row (rotate by the previous yaw and translate,
// then rotate back by the current yaw); with a fill value torchvision samples an appended ones channel with the same
// grid and multiplies (its _apply_grid_transform), so pixels whose footprint is partly outside the image are
// attenuated a second time.  Both chained resamplings are kept (two launches): composing them into one would change
// the result (each resampling blurs).
//
// theta[b] = {m0..m5}: torchvision's inverse affine matrix in pixel units about the image centre; the source pixel of
// output pixel (x, y) is
//   sx = m0 X + m1 Y + m2 + (W-1)/2,   sy = m3 X + m4 Y + m5 + (H-1)/2,   X = x - W/2 + 1/2, Y = y - H/2 + 1/2
// (its affine grid followed by grid_sample(align_corners=False), with the two normalisations cancelled).
// HBM-bound: every output element is written once and reads 4 taps that neighbouring threads share through L1/L2;
// one thread owns one output pixel and walks the channels (NCHW: consecutive lanes = consecutive x).
#include "bevr_common.h"

namespace {

struct WarpTaps {
  int x0, y0;
  float w00, w01, w10, w11;   // in-bounds weights (0 for a tap outside the image)
  float mask;                 // their sum: the resampled ones channel
};

__device__ __forceinline__ WarpTaps warp_taps(const float* th, int x, int y, int H, int W) {
  const float X = (float)x - 0.5f * (float)W + 0.5f, Y = (float)y - 0.5f * (float)H + 0.5f;
  float sx = th[0] * X + th[1] * Y + th[2] + 0.5f * (float)(W - 1);
  float sy = th[3] * X + th[4] * Y + th[5] + 0.5f * (float)(H - 1);
  // clamp before the int conversion so NaN / huge poses cannot index out of range (every tap is then outside)
  sx = fminf(fmaxf(sx, -2.0f), (float)W + 1.0f);
  sy = fminf(fmaxf(sy, -2.0f), (float)H + 1.0f);
  const float x0f = floorf(sx), y0f = floorf(sy);
  const float fx = sx - x0f, fy = sy - y0f;
  WarpTaps t;
  t.x0 = (int)x0f;
  t.y0 = (int)y0f;
  const bool vx0 = t.x0 >= 0 && t.x0 < W, vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
  const bool vy0 = t.y0 >= 0 && t.y0 < H, vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
  t.w00 = (vx0 && vy0) ? (1.f - fx) * (1.f - fy) : 0.f;
  t.w01 = (vx1 && vy0) ? fx * (1.f - fy) : 0.f;
  t.w10 = (vx0 && vy1) ? (1.f - fx) * fy : 0.f;
  t.w11 = (vx1 && vy1) ? fx * fy : 0.f;
  t.mask = (t.w00 + t.w01) + (t.w10 + t.w11);
  return t;
}

template <bool BWD>
__global__ __launch_bounds__(256) void affine_warp_kernel(const float* __restrict__ src, const float* __restrict__ theta,
                                                          float* __restrict__ dst, int B, int C, int H, int W) {
  const long long total = (long long)B * H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const int b = (int)(idx / ((long long)W * H));
    const WarpTaps t = warp_taps(theta + b * 6, x, y, H, W);
    if (t.mask == 0.f) {   // every tap outside: the output is 0 and nothing flows back
      if (!BWD)
        for (int c = 0; c < C; ++c) dst[((size_t)(b * C + c) * H + y) * W + x] = 0.f;
      continue;
    }
    // tap offsets inside one channel plane; an out-of-image tap has weight 0 and is redirected to tap 0's neighbour
    // that IS inside (never dereferenced out of range)
    const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
    const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
    const int o00 = ya * W + xa, o01 = ya * W + xb, o10 = yb * W + xa, o11 = yb * W + xb;
    const float m = t.mask;
    for (int c = 0; c < C; ++c) {
      const size_t plane = (size_t)(b * C + c) * H * W;
      if (!BWD) {
        const float* s = src + plane;
        const float v = (s[o00] * t.w00 + s[o01] * t.w01) + (s[o10] * t.w10 + s[o11] * t.w11);
        dst[plane + (size_t)y * W + x] = v * m;
      } else {   // src = d(out), dst = d(img): the transposed map, scattered
        const float g = src[plane + (size_t)y * W + x] * m;
        float* dgi = dst + plane;
        if (t.w00 != 0.f) atomicAdd(dgi + o00, g * t.w00);
        if (t.w01 != 0.f) atomicAdd(dgi + o01, g * t.w01);
        if (t.w10 != 0.f) atomicAdd(dgi + o10, g * t.w10);
        if (t.w11 != 0.f) atomicAdd(dgi + o11, g * t.w11);
      }
    }
  }
}

int check(const float* a, const float* th, float* o, int B, int C, int H, int W) {
  if (!a || !th || !o) return BEVR_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (long long)B * C * H * W >= (1LL << 40)) return BEVR_E_SHAPE;
  return BEVR_OK;
}

}  // namespace

extern "C" int bevr_affine_warp_fwd(const float* img, const float* theta, float* out, int B, int C, int H, int W,
                                    void* stream) {
  const int rc = check(img, theta, out, B, C, H, W);
  if (rc) return rc;
  const long long total = (long long)B * H * W;
  const int grid = (int)min((total + 255) / 256, (long long)256 * 32);
  hipLaunchKernelGGL((affine_warp_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, theta, out, B, C, H, W);
  return (int)hipGetLastError();
}

extern "C" int bevr_affine_warp_bwd(const float* dout, const float* theta, float* dimg, int B, int C, int H, int W,
                                    void* stream) {
  const int rc = check(dout, theta, dimg, B, C, H, W);
  if (rc) return rc;
  const long long total = (long long)B * H * W;
  const int grid = (int)min((total + 255) / 256, (long long)256 * 32);
  hipLaunchKernelGGL((affine_warp_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, theta, dimg, B, C, H, W);
  return (int)hipGetLastError();
}
