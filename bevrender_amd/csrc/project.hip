// BEV pillar grid -> camera pixels: rigid inverse, pinhole projection, perspective divide,
// int-truncated in-bound mask, masked -> pixel (0,0), normalise to [-1, 1].
// Replaces BEV2CameraProjector.bev_grid_to_camera + get_in_bound_mask, model/bev_cmr_proj.py:61-113.
// Init-time and tiny (ncam * P points); one thread per (camera, point), coalesced over points.
#include "bevr_common.h"

namespace {

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ pts, const float* __restrict__ cam_inv,
                                                      const float* __restrict__ Kmat, float* __restrict__ out, int ncam,
                                                      int P, int img_w, int img_h, const uint8_t* __restrict__ ref_img,
                                                      int ref_c, int ref_h, int ref_w) {
  const int cam = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float* T = cam_inv + cam * 16;
  const float* Km = Kmat + cam * 9;
  const float x = pts[i], y = pts[P + i], z = pts[2 * P + i], w = pts[3 * P + i];
  // same association order as a row-times-column dot product; no FMA contraction so that the result
  // tracks the reference's fp32 matmul as closely as a different machine can
  float c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    c[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4 * r], x), __fmul_rn(T[4 * r + 1], y)), __fmul_rn(T[4 * r + 2], z)),
                     __fmul_rn(T[4 * r + 3], w));
  float uvw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    uvw[r] = __fadd_rn(__fadd_rn(__fmul_rn(Km[3 * r], c[0]), __fmul_rn(Km[3 * r + 1], c[1])), __fmul_rn(Km[3 * r + 2], c[2]));
  float u = __fdiv_rn(uvw[0], uvw[2]), v = __fdiv_rn(uvw[1], uvw[2]);
  // .to(torch.int32): truncation toward zero; NaN/inf are outside every bound below
  const bool finite = (u == u) && (v == v) && fabsf(u) < 2.0e9f && fabsf(v) < 2.0e9f;
  const int iu = finite ? (int)u : -1, iv = finite ? (int)v : -1;
  bool ok = finite && iv >= 0 && iv < img_h - 1 && iu >= 0 && iu < img_w - 1;
  if (ref_img) {
    // remove_ref_in_gray (model/bev_cmr_proj.py:114-122): the reference image is read at the truncated pixel of every
    // point (pixel (0, 0) for points already outside); a pixel with exactly three channels equal to 128 masks the point
    const int pu = ok ? iu : 0, pv = ok ? iv : 0;
    int n128 = 0;
    for (int ch = 0; ch < ref_c; ++ch)
      n128 += ref_img[(((size_t)cam * ref_c + ch) * ref_h + pv) * ref_w + pu] == 128;
    ok = ok && (n128 != 3);
  }
  if (!ok) { u = 0.f; v = 0.f; }
  u = __fsub_rn(__fmul_rn(__fdiv_rn(u, (float)(img_w - 1)), 2.0f), 1.0f);
  v = __fsub_rn(__fmul_rn(__fdiv_rn(v, (float)(img_h - 1)), 2.0f), 1.0f);
  out[(size_t)cam * 2 * P + i] = u;
  out[(size_t)cam * 2 * P + P + i] = v;
}

}  // namespace

extern "C" int bevr_project_bev_grid_masked(const float* points_3d, const float* cam_inv, const float* Kmat, float* out,
                                            int ncam, int P, int img_w, int img_h, const uint8_t* ref_img, int ref_c,
                                            int ref_h, int ref_w, void* stream) {
  if (!points_3d || !cam_inv || !Kmat || !out) return BEVR_E_NULL;
  if (ncam <= 0 || P <= 0 || img_w < 2 || img_h < 2) return BEVR_E_SHAPE;
  // every pixel the in-bound test admits must exist in the reference image
  if (ref_img && (ref_c <= 0 || ref_h < img_h - 1 || ref_w < img_w - 1)) return BEVR_E_SHAPE;
  dim3 grid((P + 255) / 256, ncam);
  hipLaunchKernelGGL(project_kernel, grid, dim3(256), 0, (hipStream_t)stream, points_3d, cam_inv, Kmat, out, ncam, P,
                     img_w, img_h, ref_img, ref_c, ref_h, ref_w);
  return (int)hipGetLastError();
}

extern "C" int bevr_project_bev_grid(const float* points_3d, const float* cam_inv, const float* Kmat, float* out,
                                     int ncam, int P, int img_w, int img_h, void* stream) {
  return bevr_project_bev_grid_masked(points_3d, cam_inv, Kmat, out, ncam, P, img_w, img_h, nullptr, 0, 0, 0, stream);
}
