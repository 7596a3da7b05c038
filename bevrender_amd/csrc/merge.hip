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

// ---------------------------------------------------------------------------------------------------------------------
// The same merge with the tap segment's half NOT materialised: the tap kernels hand over Rn[ph][m][12] (the softmax
// weights summed per pixel tap, normalised) and the caller the 12 pixels' value rows Vp[ph][12][32] and the value bias
// bv[heads][32]:   O_c = Rn Vp + bv   (csrc/attn_tap.h: V_n = sum_t w_t(n) Vpix_t + bv) is formed in registers on the
// way -- a (M x 12) x (12 x 32) product per (problem, head) and an add that were two more passes over the packed output,
// and in the backward two thin batched GEMMs and a row sum.  A workgroup stays inside one (problem, head): Vp sits in
// registers, the backward's dVp / dbv partial sums too (one atomic per element and workgroup at the end).
namespace {

constexpr int MT_CHUNKS = 32;      // workgroups per (problem, head)

__device__ __forceinline__ float add8(float v) {      // sum over the 8 lanes of a row (lanes 8 k .. 8 k + 7)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // ^1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // ^2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // 7 - i
  return v;
}

struct TapRow { f32x4 r0, r1, r2; };      // Rn[row][0 .. 11]
__device__ __forceinline__ f32x4 tap_oc(const TapRow& rn, const f32x4 (&vp)[12], const f32x4& bv4) {
  f32x4 o = bv4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    o += vp[t] * rn.r0[t];
    o += vp[4 + t] * rn.r1[t];
    o += vp[8 + t] * rn.r2[t];
  }
  return o;
}

__global__ __launch_bounds__(MG_THREADS) void merge_tap_fwd_kernel(MergeGeom g, const float* __restrict__ O_r,
                                                                   const float* __restrict__ L_r, const float* __restrict__ Rn,
                                                                   const float* __restrict__ L_c, const float* __restrict__ Vp,
                                                                   const float* __restrict__ bv, float* __restrict__ out) {
  const int Mp = g.S * g.Sp;
  const int bvh = blockIdx.x, hh = bvh % g.heads, bvi = bvh / g.heads;
  const int b = bvi / g.views, v = bvi - b * g.views;
  const int c4 = threadIdx.x & 7, r32 = threadIdx.x >> 3;
  f32x4 vp[12];
#pragma unroll
  for (int t = 0; t < 12; ++t) vp[t] = *reinterpret_cast<const f32x4*>(Vp + ((size_t)bvh * 12 + t) * 32 + 4 * c4);
  const f32x4 bv4 = *reinterpret_cast<const f32x4*>(bv + hh * 32 + 4 * c4);
  const int per = ((Mp / 32 + MT_CHUNKS - 1) / MT_CHUNKS) * 32, m0 = blockIdx.y * per, m1 = min(Mp, m0 + per);
  for (int mq = m0 + r32; mq < m1; mq += 32) {
    const int j = mq / g.Sp, i = mq - j * g.Sp;
    if (i >= g.S || 4 * c4 >= g.c) continue;
    const size_t row = (size_t)bvh * Mp + mq;
    const f32x4 orr = *reinterpret_cast<const f32x4*>(O_r + row * 32 + 4 * c4);
    TapRow rn;
    rn.r0 = *reinterpret_cast<const f32x4*>(Rn + row * 12);
    rn.r1 = *reinterpret_cast<const f32x4*>(Rn + row * 12 + 4);
    rn.r2 = *reinterpret_cast<const f32x4*>(Rn + row * 12 + 8);
    const f32x4 oc = tap_oc(rn, vp, bv4);
    const float lr = L_r[row], lc = L_c[row];
    const float mx = fmaxf(lr, lc);
    const float er = exp2f(lr - mx), ec = exp2f(lc - mx);
    const float inv = 1.0f / (er + ec);
    const size_t oidx = ((size_t)b * g.S * g.S + (size_t)i * g.S + j) * ((size_t)g.views * g.heads * g.c) +
                        (size_t)(v * g.heads + hh) * g.c + 4 * c4;
    *reinterpret_cast<f32x4*>(out + oidx) = orr * (er * inv) + oc * (ec * inv);
  }
}

__global__ __launch_bounds__(MG_THREADS) void merge_tap_bwd_kernel(MergeGeom g, const float* __restrict__ dout,
                                                                   const float* __restrict__ O_r, const float* __restrict__ L_r,
                                                                   const float* __restrict__ Rn, const float* __restrict__ L_c,
                                                                   const float* __restrict__ Vp, const float* __restrict__ bv,
                                                                   float* __restrict__ dO_r, float* __restrict__ dL_r,
                                                                   float* __restrict__ dRn, float* __restrict__ dL_c,
                                                                   float* __restrict__ dVp, float* __restrict__ dbv) {
  __shared__ f32x4 red[13][MG_THREADS];
  const int Mp = g.S * g.Sp;
  const int bvh = blockIdx.x, hh = bvh % g.heads, bvi = bvh / g.heads;
  const int b = bvi / g.views, v = bvi - b * g.views;
  const int c4 = threadIdx.x & 7, r32 = threadIdx.x >> 3;
  f32x4 vp[12], acc[12], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 12; ++t) {
    vp[t] = *reinterpret_cast<const f32x4*>(Vp + ((size_t)bvh * 12 + t) * 32 + 4 * c4);
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const f32x4 bv4 = *reinterpret_cast<const f32x4*>(bv + hh * 32 + 4 * c4);
  const bool c_live = 4 * c4 < g.c;
  const int per = ((Mp / 32 + MT_CHUNKS - 1) / MT_CHUNKS) * 32, m0 = blockIdx.y * per, m1 = min(Mp, m0 + per);
  // (whole rows of 8 lanes run the loop together: m0, m1 and the step are multiples of 32, Mp too)
  for (int mq = m0 + r32; mq < m1; mq += 32) {
    const int j = mq / g.Sp, i = mq - j * g.Sp;
    const bool row_live = i < g.S;
    const size_t row = (size_t)bvh * Mp + mq;
    f32x4 gq = {0.f, 0.f, 0.f, 0.f};
    if (row_live && c_live) {
      const size_t oidx = ((size_t)b * g.S * g.S + (size_t)i * g.S + j) * ((size_t)g.views * g.heads * g.c) +
                          (size_t)(v * g.heads + hh) * g.c + 4 * c4;
      gq = *reinterpret_cast<const f32x4*>(dout + oidx);
    }
    const f32x4 orr = *reinterpret_cast<const f32x4*>(O_r + row * 32 + 4 * c4);
    TapRow rn;
    rn.r0 = *reinterpret_cast<const f32x4*>(Rn + row * 12);
    rn.r1 = *reinterpret_cast<const f32x4*>(Rn + row * 12 + 4);
    rn.r2 = *reinterpret_cast<const f32x4*>(Rn + row * 12 + 8);
    const f32x4 oc = tap_oc(rn, vp, bv4);
    const float lr = L_r[row], lc = L_c[row];
    const float mx = fmaxf(lr, lc);
    const float er = exp2f(lr - mx), ec = exp2f(lc - mx);
    const float inv = 1.0f / (er + ec);
    const float ar = row_live ? er * inv : 0.f, ac = row_live ? ec * inv : 0.f;
    *reinterpret_cast<f32x4*>(dO_r + row * 32 + 4 * c4) = gq * ar;
    const f32x4 goc = gq * ac;      // gradient of O_c
    const f32x4 df = orr - oc;
    const float p = add8(gq[0] * df[0] + gq[1] * df[1] + gq[2] * df[2] + gq[3] * df[3]);
    // d Rn[t] = sum_c goc[c] Vp[t][c]: this lane's four channels, then the row's eight lanes
    float dr[12];
#pragma unroll
    for (int t = 0; t < 12; ++t)
      dr[t] = add8(goc[0] * vp[t][0] + goc[1] * vp[t][1] + goc[2] * vp[t][2] + goc[3] * vp[t][3]);
    // lane c4 writes elements c4 and (c4 < 4) 8 + c4 of the row
    float w0 = dr[0], w1 = dr[8];
#pragma unroll
    for (int t = 1; t < 8; ++t) w0 = c4 == t ? dr[t] : w0;
#pragma unroll
    for (int t = 1; t < 4; ++t) w1 = c4 == t ? dr[8 + t] : w1;
    dRn[row * 12 + c4] = w0;
    if (c4 < 4) dRn[row * 12 + 8 + c4] = w1;
    if (c4 == 0) {
      const float gl = row_live ? 0.6931471805599453f * ar * ac * p : 0.f;
      dL_r[row] = gl;
      dL_c[row] = -gl;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] += goc * rn.r0[t];
      acc[4 + t] += goc * rn.r1[t];
      acc[8 + t] += goc * rn.r2[t];
    }
    accb += goc;
  }
  // the workgroup's 32 row slots summed through LDS; one atomic per element of dVp / dbv and workgroup
#pragma unroll
  for (int t = 0; t < 12; ++t) red[t][threadIdx.x] = acc[t];
  red[12][threadIdx.x] = accb;
  __syncthreads();
  for (int e = threadIdx.x; e < 13 * 8; e += MG_THREADS) {
    const int t = e >> 3, q = e & 7;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 32; ++r) s += red[t][r * 8 + q];
    float* dst = t < 12 ? dVp + ((size_t)bvh * 12 + t) * 32 + 4 * q : dbv + hh * 32 + 4 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(dst + k, s[k]);
  }
}

}  // namespace

extern "C" int bevr_merge_tap_fwd(const float* O_r, const float* L_r, const float* Rn, const float* L_c, const float* Vp,
                                  const float* bv, float* out, int n_prob, int views, int heads, int S, int Sp, int c,
                                  void* stream) {
  const int rc = merge_check(n_prob, views, heads, S, Sp, c);
  if (rc) return rc;
  if (!O_r || !L_r || !Rn || !L_c || !Vp || !bv || !out) return BEVR_E_NULL;
  if (!bevr_aligned16(O_r) || !bevr_aligned16(Rn) || !bevr_aligned16(Vp) || !bevr_aligned16(bv) || !bevr_aligned16(out))
    return BEVR_E_ALIGN;
  const MergeGeom g = {n_prob, views, heads, S, Sp, c};
  hipLaunchKernelGGL(merge_tap_fwd_kernel, dim3(n_prob * heads, MT_CHUNKS), dim3(MG_THREADS), 0, (hipStream_t)stream, g, O_r,
                     L_r, Rn, L_c, Vp, bv, out);
  return (int)hipGetLastError();
}

extern "C" int bevr_merge_tap_bwd(const float* dout, const float* O_r, const float* L_r, const float* Rn, const float* L_c,
                                  const float* Vp, const float* bv, float* dO_r, float* dL_r, float* dRn, float* dL_c,
                                  float* dVp, float* dbv, int n_prob, int views, int heads, int S, int Sp, int c, void* stream) {
  const int rc = merge_check(n_prob, views, heads, S, Sp, c);
  if (rc) return rc;
  if (!dout || !O_r || !L_r || !Rn || !L_c || !Vp || !bv || !dO_r || !dL_r || !dRn || !dL_c || !dVp || !dbv) return BEVR_E_NULL;
  if (!bevr_aligned16(dout) || !bevr_aligned16(O_r) || !bevr_aligned16(Rn) || !bevr_aligned16(Vp) || !bevr_aligned16(bv) ||
      !bevr_aligned16(dO_r))
    return BEVR_E_ALIGN;
  const MergeGeom g = {n_prob, views, heads, S, Sp, c};
  hipLaunchKernelGGL(merge_tap_bwd_kernel, dim3(n_prob * heads, MT_CHUNKS), dim3(MG_THREADS), 0, (hipStream_t)stream, g, dout,
                     O_r, L_r, Rn, L_c, Vp, bv, dO_r, dL_r, dRn, dL_c, dVp, dbv);
  return (int)hipGetLastError();
}
