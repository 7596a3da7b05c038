// What every 16-bit attention backward needs from the cotangent before its first kernel, in ONE pass over dO and O:
//   dOe[ph][m][32]   the cotangent rows rounded to the operand type E (bf16 / fp16), times an optional power-of-two scale
//   dOt[ph][32][Mp]  the same transposed, with the in-32 permutation over m the kernels' transposed operands use
//                    (position p holds row 32 (p / 32) + perm32(p % 32), perm32 = bits 2 and 3 swapped: ops._perm_t)
//   delta[ph][m]     rowsum(dOe o O) - log2(e) scale dLSE  -- from the ROUNDED cotangent, the one the kernels contract with V
//                    for dP: dS = P (dP - delta) then cancels where one key holds the row (ops._AttnCore.backward)
//   stats[0], [1]    max_m ||dOe_m||^2 and max_m |delta_m| (atomic max on the bit patterns; the caller zeroes them): the
//                    bound of |dP - delta| behind the backward's fixed-point scale
// Replaces, per backward call, a cast, a cast back, a product, a row sum, a row norm, an index_select and a transposing
// copy over the (B V, h, Mp, 32) cotangent (reference: autograd through model/SCA_deform_attn.py:331-413): 2.5 ms per SCA
// call at the benchmark shape, 1.65 GB of compulsory traffic.
#include "bevr_common.h"

namespace {

constexpr int BP_WAVES = 4;
constexpr int BP_PITCH = 34;      // 16-bit elements per staged row: 17 dwords, odd -- the transposed reads spread over the banks

template <int PREC>
__global__ __launch_bounds__(64 * BP_WAVES) void attn_bwd_prep_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ dLSE,
                                                                     const float* __restrict__ LSE0, uint16_t* __restrict__ dOe,
                                                                     uint16_t* __restrict__ dOt, float* __restrict__ delta,
                                                                     unsigned* __restrict__ stats, long long n_blk, int Mp) {
  __shared__ uint16_t tile[BP_WAVES][32 * BP_PITCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r8 = lane >> 3, c4 = lane & 7;
  const float sc = scale ? *scale : 1.0f;
  float mx_n2 = 0.f, mx_d = 0.f;
  uint16_t* tl = tile[wave];
  for (long long blk = (long long)blockIdx.x * BP_WAVES + wave; blk < n_blk; blk += (long long)gridDim.x * BP_WAVES) {
    const long long row0 = blk * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long row = row0 + 8 * k + r8;
      const f32x4 g = *reinterpret_cast<const f32x4*>(dO + row * 32 + 4 * c4) * sc;
      const f32x4 o = *reinterpret_cast<const f32x4*>(O + row * 32 + 4 * c4);
      uint2 w;
      w.x = Half<PREC>::pack2(g[0], g[1]);
      w.y = Half<PREC>::pack2(g[2], g[3]);
      const float e0 = Half<PREC>::lo(w.x), e1 = Half<PREC>::hi(w.x), e2 = Half<PREC>::lo(w.y), e3 = Half<PREC>::hi(w.y);
      float d = e0 * o[0] + e1 * o[1] + e2 * o[2] + e3 * o[3];
      float n2 = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
#pragma unroll
      for (int sft = 1; sft < 8; sft <<= 1) {
        d += __shfl_xor(d, sft);
        n2 += __shfl_xor(n2, sft);
      }
      *reinterpret_cast<uint2*>(dOe + row * 32 + 4 * c4) = w;
      uint32_t* ts = reinterpret_cast<uint32_t*>(tl + (8 * k + r8) * BP_PITCH + 4 * c4);      // 4-byte aligned: even pitch
      ts[0] = w.x;
      ts[1] = w.y;
      if (c4 == 0) {
        if (dLSE) {
          const float l0 = LSE0[row];
          const bool fin = l0 - l0 == 0.f;      // finite
          d -= fin ? dLSE[row] * 1.4426950408889634f * sc : 0.f;
        }
        delta[row] = d;
        mx_n2 = fmaxf(mx_n2, n2);
        mx_d = fmaxf(mx_d, fabsf(d));
      }
    }
    // the wave's own tile: no barrier, the LDS traffic of one wave is ordered
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const int c = lane & 31, half = lane >> 5;
      u32x4 lo4, hi4;
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        const int p0 = 16 * half + t, p1 = p0 + 1;
        const int q0 = (p0 & ~12) | ((p0 & 4) << 1) | ((p0 & 8) >> 1), q1 = (p1 & ~12) | ((p1 & 4) << 1) | ((p1 & 8) >> 1);
        const uint32_t pr = (uint32_t)tl[q0 * BP_PITCH + c] | ((uint32_t)tl[q1 * BP_PITCH + c] << 16);
        if (t < 8) lo4[t >> 1] = pr; else hi4[(t - 8) >> 1] = pr;
      }
      const long long ph = row0 / Mp;
      const long long m0 = row0 - ph * Mp;
      uint16_t* dst = dOt + (ph * 32 + c) * (long long)Mp + m0 + 16 * half;
      *reinterpret_cast<u32x4*>(dst) = lo4;
      *reinterpret_cast<u32x4*>(dst + 8) = hi4;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile is read before the next block overwrites it
  }
  // non-negative floats order like their bit patterns
#pragma unroll
  for (int sft = 8; sft < 64; sft <<= 1) {
    mx_n2 = fmaxf(mx_n2, __shfl_xor(mx_n2, sft));
    mx_d = fmaxf(mx_d, __shfl_xor(mx_d, sft));
  }
  if (lane == 0) {
    atomicMax(stats, __builtin_bit_cast(unsigned, mx_n2));
    atomicMax(stats + 1, __builtin_bit_cast(unsigned, mx_d));
  }
}

}  // namespace

extern "C" int bevr_attn_bwd_prep(const float* dO, const float* O, const float* scale, const float* dLSE, const float* LSE0,
                                  void* dOe, void* dOt, float* delta, float* stats, int n_ph, int Mp, int precision,
                                  void* stream) {
  if (!dO || !O || !dOe || !dOt || !delta || !stats || (dLSE && !LSE0)) return BEVR_E_NULL;
  if (n_ph <= 0 || Mp <= 0 || (Mp & 31)) return BEVR_E_SHAPE;
  if (!is16(precision)) return BEVR_E_PRECISION;
  if (!bevr_aligned16(dO) || !bevr_aligned16(O) || !bevr_aligned16(dOe) || !bevr_aligned16(dOt)) return BEVR_E_ALIGN;
  const long long n_blk = (long long)n_ph * (Mp / 32);
  const long long want = (n_blk + BP_WAVES - 1) / BP_WAVES;
  const int grid = (int)(want < 256 * 8 ? want : 256 * 8);
  hipStream_t st = (hipStream_t)stream;
  if (precision == BEVR_PREC_BF16)
    hipLaunchKernelGGL((attn_bwd_prep_kernel<BEVR_PREC_BF16>), dim3(grid), dim3(64 * BP_WAVES), 0, st, dO, O, scale, dLSE, LSE0,
                       (uint16_t*)dOe, (uint16_t*)dOt, delta, (unsigned*)stats, n_blk, Mp);
  else
    hipLaunchKernelGGL((attn_bwd_prep_kernel<BEVR_PREC_F16>), dim3(grid), dim3(64 * BP_WAVES), 0, st, dO, O, scale, dLSE, LSE0,
                       (uint16_t*)dOe, (uint16_t*)dOt, delta, (unsigned*)stats, n_blk, Mp);
  return (int)hipGetLastError();
}
