// BEV pillar grid -> camera pixels: rigid inverse, pinhole projection, perspective divide,
// int-truncated in-bound mask, masked -> pixel (0,0), normalise to [-1, 1].
// Replaces BEV2CameraProjector.bev_grid_to_camera + get_in_bound_mask, model/bev_cmr_proj.py:61-113.
// Init-time and tiny (ncam * P points); one thread per (camera, point), coalesced over points.
#include "bevr_common.h"

namespace {

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ pts, const float* __restrict__ cam_inv,
                                                      const float* __restrict__ Kmat, float* __restrict__ out, int ncam,
                                                      int P, int img_w, int img_h) {
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
  const bool ok = finite && iv >= 0 && iv < img_h - 1 && iu >= 0 && iu < img_w - 1;
  if (!ok) { u = 0.f; v = 0.f; }
  u = __fsub_rn(__fmul_rn(__fdiv_rn(u, (float)(img_w - 1)), 2.0f), 1.0f);
  v = __fsub_rn(__fmul_rn(__fdiv_rn(v, (float)(img_h - 1)), 2.0f), 1.0f);
  out[(size_t)cam * 2 * P + i] = u;
  out[(size_t)cam * 2 * P + P + i] = v;
}

}  // namespace

extern "C" int bevr_project_bev_grid(const float* points_3d, const float* cam_inv, const float* Kmat, float* out,
                                     int ncam, int P, int img_w, int img_h, void* stream) {
  if (!points_3d || !cam_inv || !Kmat || !out) return BEVR_E_NULL;
  if (ncam <= 0 || P <= 0 || img_w < 2 || img_h < 2) return BEVR_E_SHAPE;
  dim3 grid((P + 255) / 256, ncam);
  hipLaunchKernelGGL(project_kernel, grid, dim3(256), 0, (hipStream_t)stream, points_3d, cam_inv, Kmat, out, ncam, P,
                     img_w, img_h);
  return (int)hipGetLastError();
}
