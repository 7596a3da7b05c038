// The attention output on its way to proj_out: merge of the two key segments' softmax halves + the unpacking of the
// per-(problem, head) packed rows into the layout the projection contracts, in ONE pass (forward and backward).
//
//   packed:  O_x[(b V + v)][h][j Sp + i][32]  float (channels c .. 31 and rows i >= S are padding), L_x[(b V + v)][h][j Sp + i]
//            the segment's log2-sum-exp;  x = r (region / gather / slab kernels: the scattered keys), c (tap kernels: the
//            projector-pinned keys).  One softmax over both segments:
//                L = log2(2^L_r + 2^L_c),  O = 2^(L_r - L) O_r + 2^(L_c - L) O_c
//   out:     out[b][i S + j][(v h + hh) c + cc]  -- the views of a sample side by side in the channel axis, view-major:
//            what proj_out of SCA contracts (reference model/SCA_deform_attn.py:415-420: per-view outputs concatenated
//            along channels, then a 1x1 convolution V C -> C); V = 1: the (B, S S, C) rows of TSA
//            (model/TSA_deform_attn.py:325-333).
// With O_c == NULL the kernels only unpack (one segment).
//
// Before: exp2 / mul / mul / add over the packed (B V, h, Mp, 32) tensors, then a permuting copy -- five passes over
// 0.55 GB each at the benchmark shape in the forward, ten in the backward (with the row sums of the two weights'
// gradients).  HBM-bound by construction: one float4 per thread, 8 threads per packed row; the backward's row sum
// d L_r = ln2 a_r a_c sum_c dO (O_r - O_c) is a shuffle reduction over those 8 lanes.
#include "bevr_common.h"

namespace {

constexpr int MG_THREADS = 256;

struct MergeGeom { int n_prob, views, heads, S, Sp, c; };

// thread -> (packed row, 4-channel group); the output position of the row; false: padding (row i >= S or channels >= c);
// row_live: i < S
__device__ __forceinline__ bool merge_index(const MergeGeom& g, long long t, long long& row, int& c4, long long& oidx,
                                            bool& row_live) {
  row = t >> 3;
  c4 = (int)(t & 7);
  const int Mp = g.S * g.Sp;
  const long long bvh = row / Mp;
  const int mq = (int)(row - bvh * Mp);
  const int j = mq / g.Sp, i = mq - j * g.Sp;
  const int bv = (int)(bvh / g.heads), hh = (int)(bvh - (long long)bv * g.heads);
  const int b = bv / g.views, v = bv - b * g.views;
  oidx = ((long long)b * g.S * g.S + (long long)i * g.S + j) * ((long long)g.views * g.heads * g.c) +
         (long long)(v * g.heads + hh) * g.c + 4 * c4;
  row_live = i < g.S;
  return row_live && 4 * c4 < g.c;
}

template <bool TWO>
__global__ __launch_bounds__(MG_THREADS) void merge_views_fwd_kernel(MergeGeom g, const float* __restrict__ O_r,
                                                                     const float* __restrict__ L_r,
                                                                     const float* __restrict__ O_c,
                                                                     const float* __restrict__ L_c, float* __restrict__ out,
                                                                     long long n_thr) {
  for (long long t = (long long)blockIdx.x * MG_THREADS + threadIdx.x; t < n_thr; t += (long long)gridDim.x * MG_THREADS) {
    long long row, oidx;
    int c4;
    bool row_live;
    if (!merge_index(g, t, row, c4, oidx, row_live)) continue;
    f32x4 o = *reinterpret_cast<const f32x4*>(O_r + row * 32 + 4 * c4);
    if constexpr (TWO) {
      const f32x4 oc = *reinterpret_cast<const f32x4*>(O_c + row * 32 + 4 * c4);
      const float lr = L_r[row], lc = L_c[row];
      // a_r = 2^lr / (2^lr + 2^lc) without overflow: relative to the larger of the two
      const float mx = fmaxf(lr, lc);
      const float er = exp2f(lr - mx), ec = exp2f(lc - mx);
      const float inv = 1.0f / (er + ec);
      o = o * (er * inv) + oc * (ec * inv);
    }
    *reinterpret_cast<f32x4*>(out + oidx) = o;
  }
}

template <bool TWO>
__global__ __launch_bounds__(MG_THREADS) void merge_views_bwd_kernel(MergeGeom g, const float* __restrict__ dout,
                                                                     const float* __restrict__ O_r,
                                                                     const float* __restrict__ L_r,
                                                                     const float* __restrict__ O_c,
                                                                     const float* __restrict__ L_c, float* __restrict__ dO_r,
                                                                     float* __restrict__ dL_r, float* __restrict__ dO_c,
                                                                     float* __restrict__ dL_c, long long n_thr) {
  // n_thr is a multiple of 64 (8 threads per row, Mp a multiple of 32): whole waves, the 8 lanes of a row converge
  for (long long t = (long long)blockIdx.x * MG_THREADS + threadIdx.x; t < n_thr; t += (long long)gridDim.x * MG_THREADS) {
    long long row, oidx;
    int c4;
    bool row_live;
    const bool live = merge_index(g, t, row, c4, oidx, row_live);
    f32x4 gq = {0.f, 0.f, 0.f, 0.f};
    if (live) gq = *reinterpret_cast<const f32x4*>(dout + oidx);
    if constexpr (TWO) {
      const f32x4 orr = *reinterpret_cast<const f32x4*>(O_r + row * 32 + 4 * c4);
      const f32x4 occ = *reinterpret_cast<const f32x4*>(O_c + row * 32 + 4 * c4);
      const float lr = L_r[row], lc = L_c[row];
      const float mx = fmaxf(lr, lc);
      const float er = exp2f(lr - mx), ec = exp2f(lc - mx);
      const float inv = 1.0f / (er + ec);
      // (a row past the grid may carry any L, infinities included: its weights are not used)
      const float ar = row_live ? er * inv : 0.f, ac = row_live ? ec * inv : 0.f;
      *reinterpret_cast<f32x4*>(dO_r + row * 32 + 4 * c4) = gq * ar;
      *reinterpret_cast<f32x4*>(dO_c + row * 32 + 4 * c4) = gq * ac;
      const f32x4 df = orr - occ;
      float p = gq[0] * df[0] + gq[1] * df[1] + gq[2] * df[2] + gq[3] * df[3];
      p += __shfl_xor(p, 1);
      p += __shfl_xor(p, 2);
      p += __shfl_xor(p, 4);
      if (c4 == 0) {
        // d a_r / d L_r = ln2 a_r a_c = -d a_c / d L_r (a_r + a_c = 1); a padding row (L = 0 on both sides, dout = 0): 0
        const float gl = row_live ? 0.6931471805599453f * ar * ac * p : 0.f;      // (p of a padding row may be NaN)
        dL_r[row] = gl;
        dL_c[row] = -gl;
      }
    } else {
      *reinterpret_cast<f32x4*>(dO_r + row * 32 + 4 * c4) = gq;
    }
  }
}

int merge_check(int n_prob, int views, int heads, int S, int Sp, int c) {
  if (n_prob <= 0 || views <= 0 || heads <= 0 || S <= 0 || c <= 0) return BEVR_E_SHAPE;
  if (n_prob % views != 0 || Sp < S || (Sp & 31) != 0 || c > 32 || (c & 3) != 0) return BEVR_E_SHAPE;
  return BEVR_OK;
}

int merge_grid(long long n_thr) {
  const long long want = (n_thr + MG_THREADS - 1) / MG_THREADS;
  return (int)(want < 256 * 16 ? want : 256 * 16);      // grid-stride above 16 workgroups per CU
}

}  // namespace

extern "C" int bevr_merge_views_fwd(const float* O_r, const float* L_r, const float* O_c, const float* L_c, float* out,
                                    int n_prob, int views, int heads, int S, int Sp, int c, void* stream) {
  const int rc = merge_check(n_prob, views, heads, S, Sp, c);
  if (rc) return rc;
  if (!O_r || !out || (O_c && (!L_r || !L_c))) return BEVR_E_NULL;
  // 16-byte accesses: the packed rows are 128 bytes; an output row segment starts at a multiple of c floats
  if (!bevr_aligned16(O_r) || !bevr_aligned16(out) || (O_c && !bevr_aligned16(O_c))) return BEVR_E_ALIGN;
  const MergeGeom g = {n_prob, views, heads, S, Sp, c};
  const long long n_thr = (long long)n_prob * heads * S * Sp * 8;
  hipStream_t st = (hipStream_t)stream;
  if (O_c)
    hipLaunchKernelGGL((merge_views_fwd_kernel<true>), dim3(merge_grid(n_thr)), dim3(MG_THREADS), 0, st, g, O_r, L_r, O_c,
                       L_c, out, n_thr);
  else
    hipLaunchKernelGGL((merge_views_fwd_kernel<false>), dim3(merge_grid(n_thr)), dim3(MG_THREADS), 0, st, g, O_r, L_r, O_c,
                       L_c, out, n_thr);
  return (int)hipGetLastError();
}

extern "C" int bevr_merge_views_bwd(const float* dout, const float* O_r, const float* L_r, const float* O_c,
                                    const float* L_c, float* dO_r, float* dL_r, float* dO_c, float* dL_c, int n_prob,
                                    int views, int heads, int S, int Sp, int c, void* stream) {
  const int rc = merge_check(n_prob, views, heads, S, Sp, c);
  if (rc) return rc;
  if (!dout || !dO_r) return BEVR_E_NULL;
  if (O_c && (!O_r || !L_r || !L_c || !dL_r || !dO_c || !dL_c)) return BEVR_E_NULL;
  if (!bevr_aligned16(dout) || !bevr_aligned16(dO_r) || (O_c && (!bevr_aligned16(O_r) || !bevr_aligned16(O_c) ||
                                                                !bevr_aligned16(dO_c))))
    return BEVR_E_ALIGN;
  const MergeGeom g = {n_prob, views, heads, S, Sp, c};
  const long long n_thr = (long long)n_prob * heads * S * Sp * 8;
  hipStream_t st = (hipStream_t)stream;
  if (O_c)
    hipLaunchKernelGGL((merge_views_bwd_kernel<true>), dim3(merge_grid(n_thr)), dim3(MG_THREADS), 0, st, g, dout, O_r, L_r,
                       O_c, L_c, dO_r, dL_r, dO_c, dL_c, n_thr);
  else
    hipLaunchKernelGGL((merge_views_bwd_kernel<false>), dim3(merge_grid(n_thr)), dim3(MG_THREADS), 0, st, g, dout, O_r, L_r,
                       O_c, L_c, dO_r, dL_r, dO_c, dL_c, n_thr);
  return (int)hipGetLastError();
}
