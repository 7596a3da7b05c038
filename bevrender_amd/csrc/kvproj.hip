// K | V operands straight from the feature map: bilinear sampling -> proj_k | proj_v (one 1x1 GEMM on the matrix
// cores) -> the attention kernels' packed per-head layouts, in one pass.  The sampled features (B', N, C) and the
// projected rows (B', N, 2C) never reach HBM, and there is no separate packing pass.
//
// Replaces, for the 16-bit operand modes, the chain  F.grid_sample (model/SCA_deform_attn.py:290-301,
// model/TSA_deform_attn.py:210-217) -> proj_k / proj_v (SCA_deform_attn.py:312-321, TSA_deform_attn.py:226-236) ->
// reshape into per-head operands, i.e. bevr_sample_fwd -> rocBLAS -> bevr_pack_kv.  The GEMM's operands are the E-rounded
// sampled features and weights (E = bf16 or fp16, f32 accumulation): the outputs are rounded to E anyway, the extra
// rounding is of the same size.  The backward keeps the unfused form (bevr_unpack_dkv -> GEMMs -> bevr_sample_bwd; it
// needs the float sampled features for the weight gradient and recomputes them with bevr_sample_fwd).
//
// One workgroup = 64 consecutive keys of one problem; 4 waves.
//   phase 1: 256 threads sample the 64 x C tile (4 channels per thread and tap, as sample.hip) into LDS as E rows;
//   phase 2: wave w takes the (kind, head) output tiles w, w + 4, ...: 32 output channels x 64 keys, contraction C, as
//            v_mfma_f32_32x32x16 in BOTH orientations -- D[out][key] (lane = key: the row layout X[n][32]) and D[key][out]
//            (lane = channel: the transposed layout Xt[ch][n], whose in-32 key order IS the accumulator's row order, see
//            bevr_common.h) -- instead of a transpose through LDS; the matrix work is ~0.1 ms per SCA call either way.
#include "bevr_common.h"

namespace {

constexpr int KVP_KEYS = 64, KVP_THREADS = 256;

struct bf16_bits { unsigned short u; };
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16_bits* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
               __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
}

struct KvGeom {
  int nb, Hi, Wi, C, N, Np, heads, c;
  long long pos_pstride;   // keys between two position blocks (the caller's array may hold more keys than N)
  int groups;              // channel groups: group gi's C / groups channels are sampled at position block b * groups + gi
                           // (model/SCA_deform_attn.py:290-301: x.reshape(B g, C / g, Hi, Wi) against pos (B g, ...))
};

template <int PREC, typename T>
__global__ __launch_bounds__(KVP_THREADS) void kv_project_kernel(KvGeom g, const T* __restrict__ feat,
                                                                 const float* __restrict__ pos,
                                                                 const uint32_t* __restrict__ Wkv,   // [2C][C] E, as dword pairs
                                                                 const float* __restrict__ bkv,      // [2C] or null
                                                                 char* __restrict__ Kr, char* __restrict__ Vr,
                                                                 char* __restrict__ Kt, char* __restrict__ Vt,
                                                                 unsigned* __restrict__ vnorm2_max,
                                                                 unsigned* __restrict__ knorm2_max) {   // [nb][heads] or null
  extern __shared__ __attribute__((aligned(16))) char xs[];   // [64 keys][C] E, row stride C * 2 + 16
  const int C = g.C, XS = C * 2 + 16;
  const int b = blockIdx.y, n0 = blockIdx.x * KVP_KEYS, tid = threadIdx.x;
  const int c4n = C >> 2;

  // ---- phase 1: sample (grid_sample bilinear, align_corners = True, zero padding) ----
  const T* fimg = feat + (size_t)b * g.Hi * g.Wi * C;
  for (int it = tid; it < KVP_KEYS * c4n; it += KVP_THREADS) {
    const int key = it / c4n, c4 = it - key * c4n;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (n0 + key < g.N) {
      const int gi = (c4 * 4) / (C / g.groups);
      const f32x2 p = *reinterpret_cast<const f32x2*>(pos + (((size_t)b * g.groups + gi) * g.pos_pstride + n0 + key) * 2);
      const float ix = (p[1] + 1.0f) * 0.5f * (float)(g.Wi - 1), iy = (p[0] + 1.0f) * 0.5f * (float)(g.Hi - 1);
      float x0f = floorf(ix), y0f = floorf(iy);
      const float fx = ix - x0f, fy = iy - y0f;
      x0f = fminf(fmaxf(x0f, -2.0f), (float)g.Wi);   // NaN / huge positions cannot index out of range
      y0f = fminf(fmaxf(y0f, -2.0f), (float)g.Hi);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const bool vx0 = x0 >= 0 && x0 < g.Wi, vx1 = x0 + 1 >= 0 && x0 + 1 < g.Wi;
      const bool vy0 = y0 >= 0 && y0 < g.Hi, vy1 = y0 + 1 >= 0 && y0 + 1 < g.Hi;
      const T* fb = fimg + c4 * 4;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 v00 = (vy0 && vx0) ? ld4(fb + ((size_t)y0 * g.Wi + x0) * C) : z;
      const f32x4 v01 = (vy0 && vx1) ? ld4(fb + ((size_t)y0 * g.Wi + x0 + 1) * C) : z;
      const f32x4 v10 = (vy1 && vx0) ? ld4(fb + ((size_t)(y0 + 1) * g.Wi + x0) * C) : z;
      const f32x4 v11 = (vy1 && vx1) ? ld4(fb + ((size_t)(y0 + 1) * g.Wi + x0 + 1) * C) : z;
      // the same expression as sample_fwd_kernel: the float samples the backward recomputes are these, bit for bit
      r = v00 * ((1.f - fx) * (1.f - fy)) + v01 * (fx * (1.f - fy)) + v10 * ((1.f - fx) * fy) + v11 * (fx * fy);
    }
    uint2 w;
    w.x = Half<PREC>::pack2(r[0], r[1]);
    w.y = Half<PREC>::pack2(r[2], r[3]);
    *reinterpret_cast<uint2*>(xs + key * XS + c4 * 8) = w;
  }
  __syncthreads();

  // ---- phase 2: the projection, per (kind, head) tile of 32 output channels ----
  const int wave = tid >> 6, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const int ksteps = C >> 4;
  for (int nt = wave; nt < 2 * g.heads; nt += KVP_THREADS / 64) {
    const int kind = nt / g.heads, head = nt - kind * g.heads;
    const bool row_ok = lq < g.c;                       // output channel lq of this head exists (head_dim c <= 32)
    const int o = kind * C + head * g.c + (row_ok ? lq : 0);
    f32x16 accR[2], accT[2];                            // [key tile]: D[out][key] and D[key][out]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accR[t][r] = 0.f; accT[t][r] = 0.f; }
    for (int s = 0; s < ksteps; ++s) {
      // element j of k-step s <-> input channel 16 s + 8 hi + j, on both operands
      u32x4 wv = *reinterpret_cast<const u32x4*>(Wkv + ((size_t)o * C + 16 * s + 8 * hi) / 2);
      if (!row_ok) wv = u32x4{0u, 0u, 0u, 0u};
      const bf16x8 wf = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xs + (t * 32 + lq) * XS + (16 * s + 8 * hi) * 2));
        accR[t] = Half<PREC>::mfma(wf, xf, accR[t]);   // rows = output channels, lane = key
        accT[t] = Half<PREC>::mfma(xf, wf, accT[t]);   // rows = keys, lane = output channel
      }
    }
    const float bias_lane = (bkv && row_ok) ? bkv[o] : 0.f;
    float vn2 = 0.f;   // largest squared row norm seen by this lane (V: the backward's scale bound; K: the forward's)
    char* Xr = kind ? Vr : Kr;
    char* Xt = kind ? Vt : Kt;
    const size_t ph = (size_t)b * g.heads + head;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // row layout X[ph][Np][32]: this lane's key, channels crow(r, hi) = 8 (r >> 2) + 4 hi + (r & 3): 4 runs of 4
      const int key = n0 + t * 32 + lq;
      const bool live = key < g.N;
      char* dst = Xr + ((ph * g.Np + key) * 32) * 2;
      float rn2 = 0.f;   // this lane's 16 channels of the row
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        float vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ch = 8 * q4 + 4 * hi + k;
          const float bo = (bkv && ch < g.c) ? bkv[kind * C + head * g.c + ch] : 0.f;
          vv[k] = live ? accR[t][4 * q4 + k] + bo : 0.f;
        }
        rn2 += vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2] + vv[3] * vv[3];
        uint2 w;
        w.x = Half<PREC>::pack2(vv[0], vv[1]);
        w.y = Half<PREC>::pack2(vv[2], vv[3]);
        *reinterpret_cast<uint2*>(dst + (8 * q4 + 4 * hi) * 2) = w;
      }
      vn2 = fmaxf(vn2, rn2 + __shfl_xor(rn2, 32));   // the other 16 channels sit in the other lane half
      // transposed layout Xt[ph][32][Np]: this lane's channel, keys crow(r, hi) of the tile = positions 8 hi .. + 7
      // (r = 0..7) and 16 + 8 hi .. + 7 (r = 8..15) of the 32-block in its perm32 order
      if (Xt) {
        char* dt = Xt + ((ph * 32 + lq) * g.Np + n0 + t * 32) * 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32x4 w;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r0 = 8 * half + 2 * k;
            const bool l0 = n0 + t * 32 + crow(r0, hi) < g.N, l1 = n0 + t * 32 + crow(r0 + 1, hi) < g.N;
            w[k] = Half<PREC>::pack2(l0 ? accT[t][r0] + bias_lane : 0.f, l1 ? accT[t][r0 + 1] + bias_lane : 0.f);
          }
          *reinterpret_cast<u32x4*>(dt + (16 * half + 8 * hi) * 2) = w;
        }
      }
    }
    // V tiles: the launch's largest squared row norm; K tiles: the (problem, head)'s (the forward's softmax reference)
    unsigned* nmax = kind ? vnorm2_max : (knorm2_max ? knorm2_max + ph : nullptr);
    if (nmax) {   // non-negative floats order like their bit patterns
#pragma unroll
      for (int sh = 16; sh > 0; sh >>= 1) vn2 = fmaxf(vn2, __shfl_xor(vn2, sh));
      if (lane == 0) atomicMax(nmax, __builtin_bit_cast(unsigned, vn2));
    }
  }
}

template <int PREC, typename T>
int launch(const KvGeom& g, const T* feat, const float* pos, const void* Wkv, const float* bkv, void* Kr, void* Vr,
           void* Kt, void* Vt, float* vn, float* kn, hipStream_t st) {
  const size_t lds = (size_t)KVP_KEYS * (g.C * 2 + 16);
  hipLaunchKernelGGL((kv_project_kernel<PREC, T>), dim3(g.Np / KVP_KEYS, g.nb), dim3(KVP_THREADS), lds, st, g, feat, pos,
                     (const uint32_t*)Wkv, bkv, (char*)Kr, (char*)Vr, (char*)Kt, (char*)Vt, (unsigned*)vn, (unsigned*)kn);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_kv_project(const void* feat, int feat_bf16, const float* pos, long long pos_pstride, const void* Wkv,
                               const float* bkv, int nb, int Hi, int Wi, int C, int N, int Np, int heads, int c,
                               int precision, void* Kr, void* Vr, void* Kt, void* Vt, float* vnorm2_max,
                               float* knorm2_max, int groups, void* stream) {
  if (!feat || !pos || !Wkv || !Kr || !Vr || !Vt) return BEVR_E_NULL;
  if (nb <= 0 || Hi < 2 || Wi < 2 || N <= 0 || Np < N || Np % KVP_KEYS || heads <= 0 || c <= 0 || c > 32 ||
      C != heads * c || (C & 15) || C > 256 || pos_pstride < N || groups <= 0 || C % groups || ((C / groups) & 3))
    return BEVR_E_SHAPE;
  if (!is16(precision)) return BEVR_E_PRECISION;   // the f32-layout modes keep the unfused chain
  if (!bevr_aligned16(feat) || !bevr_aligned16(Wkv) || !bevr_aligned16(Kr) || !bevr_aligned16(Vr) || !bevr_aligned16(Vt) ||
      (Kt && !bevr_aligned16(Kt)) || (reinterpret_cast<uintptr_t>(pos) & 7))
    return BEVR_E_ALIGN;
  const KvGeom g{nb, Hi, Wi, C, N, Np, heads, c, pos_pstride, groups};
  hipStream_t st = (hipStream_t)stream;
  if (precision == BEVR_PREC_BF16)
    return feat_bf16 ? launch<BEVR_PREC_BF16>(g, static_cast<const bf16_bits*>(feat), pos, Wkv, bkv, Kr, Vr, Kt, Vt, vnorm2_max, knorm2_max, st)
                     : launch<BEVR_PREC_BF16>(g, static_cast<const float*>(feat), pos, Wkv, bkv, Kr, Vr, Kt, Vt, vnorm2_max, knorm2_max, st);
  return feat_bf16 ? launch<BEVR_PREC_F16>(g, static_cast<const bf16_bits*>(feat), pos, Wkv, bkv, Kr, Vr, Kt, Vt, vnorm2_max, knorm2_max, st)
                   : launch<BEVR_PREC_F16>(g, static_cast<const float*>(feat), pos, Wkv, bkv, Kr, Vr, Kt, Vt, vnorm2_max, knorm2_max, st);
}
