// Key positions from the offset heads' outputs, in the attention's key order, in one pass -- and the adjoint.
//
// Reference: model/SCA_deform_attn.py:248-277 (the offset head's (b g) d (h n) w output: even BEV rows -> y-offset of
// key row h, odd rows -> x-offset, key column w * D + d; tanh * range, + reference point, or clamp) and
// model/TSA_deform_attn.py:170-196 (offset (y, x) per key-grid pixel, tanh * range, + regular grid).  The stock-op chain
// for this is a reshape / permute copy, tanh, two multiplies, a permute copy and an add per view, a stack over the
// views and the gather into the static k-d key order: ~10 launches per view forward, more backward, each over the
// same 2 N floats.  Here: pos[b][v][gi][n'] = f(off[v][b G + gi][src(order[v][n'])]) + ref[v][order[v][n']], one launch.
#include "bevr_common.h"

namespace {

struct KeyPosGeom {
  int V, P, N;        // views, problems per view (B * G), keys per problem
  int G;              // channel groups: problem p = b * G + gi; output block (b * V + v) * G + gi
  int sca;            // 1: SCA row-split indexing with S, D; 0: TSA (off is [N][2])
  int S, D;           // SCA: BEV side and depth bins (Hk = S / 2, Wk = S * D)
  int use_tanh;       // 1: tanh(off) * scale + ref; 0: clamp(off + ref, -1, 1)
  float sy, sx;       // range * factor per component
};

// element index of component c (0 = y, 1 = x) of key n inside one (view, problem) offset block
__device__ __forceinline__ size_t off_index(const KeyPosGeom& g, int n, int c) {
  if (!g.sca) return (size_t)n * 2 + c;
  const int Wk = g.S * g.D;
  const int hk = n / Wk, wk = n - hk * Wk;
  const int w = wk / g.D, d = wk - w * g.D;
  return ((size_t)(2 * hk + c) * g.S + w) * g.D + d;
}

__global__ __launch_bounds__(256) void keypos_fwd_kernel(KeyPosGeom g, const float* __restrict__ off,
                                                         const float* __restrict__ ref, const int* __restrict__ order,
                                                         float* __restrict__ pos) {
  const size_t per_view = (size_t)g.P * g.N;
  const size_t total = per_view * g.V;
  const size_t block = g.sca ? (size_t)g.S * g.S * g.D : (size_t)g.N * 2;   // floats of one (view, problem) offset block
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    // idx = ((b * V + v) * G + gi) * N + n'   (output layout [B][V][G][N][2])
    const int n2 = (int)(idx % g.N);
    const size_t bvg = idx / g.N;
    const int gi = (int)(bvg % g.G), v = (int)((bvg / g.G) % g.V), p = (int)(bvg / ((size_t)g.G * g.V)) * g.G + gi;
    const int n = order ? order[(size_t)v * g.N + n2] : n2;
    const float* ob = off + ((size_t)v * g.P + p) * block;
    const float oy = ob[off_index(g, n, 0)], ox = ob[off_index(g, n, 1)];
    const f32x2 r = *reinterpret_cast<const f32x2*>(ref + ((size_t)v * g.N + n) * 2);
    f32x2 o;
    if (g.use_tanh) {
      o[0] = tanhf(oy) * g.sy + r[0];
      o[1] = tanhf(ox) * g.sx + r[1];
    } else {
      o[0] = fminf(fmaxf(oy + r[0], -1.0f), 1.0f);
      o[1] = fminf(fmaxf(ox + r[1], -1.0f), 1.0f);
    }
    *reinterpret_cast<f32x2*>(pos + idx * 2) = o;
  }
}

// adjoint: every offset element belongs to exactly one key (the order is a permutation), so d(off) is written, not
// accumulated.  The local derivative is recomputed from off (tanh) or from off + ref (clamp: 1 inside the range).
__global__ __launch_bounds__(256) void keypos_bwd_kernel(KeyPosGeom g, const float* __restrict__ off,
                                                         const float* __restrict__ ref, const int* __restrict__ order,
                                                         const float* __restrict__ dpos, float* __restrict__ doff) {
  const size_t per_view = (size_t)g.P * g.N;
  const size_t total = per_view * g.V;
  const size_t block = g.sca ? (size_t)g.S * g.S * g.D : (size_t)g.N * 2;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n2 = (int)(idx % g.N);
    const size_t bvg = idx / g.N;
    const int gi = (int)(bvg % g.G), v = (int)((bvg / g.G) % g.V), p = (int)(bvg / ((size_t)g.G * g.V)) * g.G + gi;
    const int n = order ? order[(size_t)v * g.N + n2] : n2;
    const size_t base = ((size_t)v * g.P + p) * block;
    const size_t iy = base + off_index(g, n, 0), ix = base + off_index(g, n, 1);
    const f32x2 gp = *reinterpret_cast<const f32x2*>(dpos + idx * 2);
    const float oy = off[iy], ox = off[ix];
    if (g.use_tanh) {
      const float ty = tanhf(oy), tx = tanhf(ox);
      doff[iy] = gp[0] * g.sy * (1.0f - ty * ty);
      doff[ix] = gp[1] * g.sx * (1.0f - tx * tx);
    } else {
      const f32x2 r = *reinterpret_cast<const f32x2*>(ref + ((size_t)v * g.N + n) * 2);
      const float py = oy + r[0], px = ox + r[1];
      doff[iy] = (py >= -1.0f && py <= 1.0f) ? gp[0] : 0.f;
      doff[ix] = (px >= -1.0f && px <= 1.0f) ? gp[1] : 0.f;
    }
  }
}

int check(int V, int P, int G, int N, int sca, int S, int D) {
  if (V <= 0 || P <= 0 || N <= 0 || G <= 0 || P % G) return BEVR_E_SHAPE;
  if (sca && (S < 2 || (S & 1) || D <= 0 || (long long)(S / 2) * S * D != N)) return BEVR_E_SHAPE;
  return BEVR_OK;
}

int grid_of(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : g);
}

}  // namespace

extern "C" int bevr_key_positions_fwd(const float* off, const float* ref, const int* order, float* pos, int V, int P,
                                      int G, int N, int sca, int S, int D, int use_tanh, float sy, float sx, void* stream) {
  if (!off || !ref || !pos) return BEVR_E_NULL;
  int rc = check(V, P, G, N, sca, S, D);
  if (rc) return rc;
  if ((reinterpret_cast<uintptr_t>(ref) & 7) || (reinterpret_cast<uintptr_t>(pos) & 7)) return BEVR_E_ALIGN;
  KeyPosGeom g{V, P, N, G, sca, S, D, use_tanh, sy, sx};
  hipLaunchKernelGGL(keypos_fwd_kernel, dim3(grid_of((size_t)V * P * N)), dim3(256), 0, (hipStream_t)stream, g, off, ref,
                     order, pos);
  return (int)hipGetLastError();
}

extern "C" int bevr_key_positions_bwd(const float* off, const float* ref, const int* order, const float* dpos,
                                      float* doff, int V, int P, int G, int N, int sca, int S, int D, int use_tanh, float sy,
                                      float sx, void* stream) {
  if (!off || !ref || !dpos || !doff) return BEVR_E_NULL;
  int rc = check(V, P, G, N, sca, S, D);
  if (rc) return rc;
  if ((reinterpret_cast<uintptr_t>(ref) & 7) || (reinterpret_cast<uintptr_t>(dpos) & 7)) return BEVR_E_ALIGN;
  KeyPosGeom g{V, P, N, G, sca, S, D, use_tanh, sy, sx};
  hipLaunchKernelGGL(keypos_bwd_kernel, dim3(grid_of((size_t)V * P * N)), dim3(256), 0, (hipStream_t)stream, g, off, ref,
                     order, dpos, doff);
  return (int)hipGetLastError();
}
