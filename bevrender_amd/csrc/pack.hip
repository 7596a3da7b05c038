// Operand packing for the attention kernels, and its transpose for the key-side gradients.
//
// The K/V projections of the sampled features come out of rocBLAS as rows (B', N, C) of float (one GEMM emits K | V
// side by side, so a row stride is taken); the attention kernels read
//   row layout        X [B'][h][Np][32] E      (E = bf16 or float; head_dim c <= 32 zero padded, keys N..Np-1 zero)
//   transposed layout Xt[B'][h][32][Np] E      with bits 2 <-> 3 of the in-32 key index swapped (bevr_common.h: the
//                                               order in which an MFMA accumulator tile is consumed as a B operand)
// The stock-op chain for this (reshape/permute/pad copy, dtype copy, index_select, transpose copy, per tensor) moved
// ~12 GB per SCA call at B = 8 for 2.5 GB of input; one pass here reads the rows once and writes each layout once.
// Reference: the reshapes of model/SCA_deform_attn.py:312-321 (proj_k / proj_v outputs -> (B h, c, N)).
#include "bevr_common.h"

namespace {

constexpr int PK = 64;   // keys per workgroup tile

__device__ __forceinline__ int perm32(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <typename E> __device__ __forceinline__ E to_elem(float x);
template <> __device__ __forceinline__ float to_elem<float>(float x) { return x; }
template <> __device__ __forceinline__ unsigned short to_elem<unsigned short>(float x) {
  return (unsigned short)(pack_bf16x2(x, 0.f) & 0xffffu);   // round to nearest even, as torch's .to(bfloat16)
}
struct half_bits { unsigned short u; };   // fp16 element (same size and layouts as bf16; distinct type for to_elem)
template <> __device__ __forceinline__ half_bits to_elem<half_bits>(float x) {
  return half_bits{(unsigned short)(Half<BEVR_PREC_F16>::pack2(x, 0.f) & 0xffffu)};   // nearest even, as .to(float16)
}

// One workgroup = 64 consecutive keys of one problem, all heads, K and V.
// SPLIT (BEVR_PREC_BF16X3; E = float): the same strides and addresses, but every 128 bytes hold the hi and lo bf16 planes
// of their 32 values in the order the split-mode fragments read them (bevr_common.h, Frag<BEVR_PREC_BF16X3>).
__device__ __forceinline__ void split_bf16(float x, unsigned short& h, unsigned short& l) {
  const unsigned hb = pack_bf16x2(x, 0.f) & 0xffffu;
  h = (unsigned short)hb;
  l = (unsigned short)(pack_bf16x2(x - __builtin_bit_cast(float, hb << 16), 0.f) & 0xffffu);
}

template <typename E, bool SPLIT = false>
__global__ __launch_bounds__(256) void pack_kv_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                      long long ld, long long pstride, int N, int Np, int heads, int c,
                                                      E* __restrict__ Kr, E* __restrict__ Vr, E* __restrict__ Kt,
                                                      E* __restrict__ Vt) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  E* tile = reinterpret_cast<E*>(smem_raw);   // [kind 2][head][key 64][32]
  // heads are processed HG at a time (blockIdx.z) so that the tile fits 64 KB of LDS for any head count
  const int b = blockIdx.y, n0 = blockIdx.x * PK, tid = threadIdx.x;
  const int hg = (int)(32768 / (2 * PK * 32 * sizeof(E)));   // 4 heads of bf16, 2 of float
  const int head0 = blockIdx.z * hg, all_heads = heads;
  const float* koff = k + (size_t)head0 * c;
  const float* voff = v + (size_t)head0 * c;
  heads = min(hg, all_heads - head0);
  const int C = heads * c;
  const int n_el = 2 * heads * PK * 32;
  if (c < 32 || n0 + PK > N) {   // zero padding (channels c..31, keys past N)
    for (int i = tid; i < n_el; i += 256) tile[i] = to_elem<E>(0.f);
    __syncthreads();
  }
  const float* src[2] = {koff + (size_t)b * pstride * ld, voff + (size_t)b * pstride * ld};
#pragma unroll
  for (int kind = 0; kind < 2; ++kind) {
    for (int i = tid; i < PK * C; i += 256) {
      const int key = i / C, ch = i - key * C;
      if (n0 + key < N) {
        const float x = src[kind][(size_t)(n0 + key) * ld + ch];
        tile[((kind * heads + ch / c) * PK + key) * 32 + ch % c] = to_elem<E>(x);
      }
    }
  }
  __syncthreads();
  if constexpr (SPLIT) {
    // row layout: per (row, 16-element half): hi(e0..7) | hi(e8..15) | lo(e0..7) | lo(e8..15); one thread per 8 elements
    for (int i = tid; i < 2 * heads * PK * 4; i += 256) {
      const int g8 = i & 3, key = (i >> 2) % PK, hk = i / (4 * PK);
      const int kind = hk / heads, head = hk - kind * heads;
      const E* srcp = tile + (hk * PK + key) * 32 + g8 * 8;
      unsigned short hb[8], lb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split_bf16(srcp[e], hb[e], lb[e]);
      char* dst = reinterpret_cast<char*>((kind ? Vr : Kr) + (((size_t)b * all_heads + head0 + head) * Np + n0 + key) * 32) +
                  (g8 >> 1) * 64 + (g8 & 1) * 16;
      u32x4 w;
      __builtin_memcpy(&w, hb, 16);
      *reinterpret_cast<u32x4*>(dst) = w;
      __builtin_memcpy(&w, lb, 16);
      *reinterpret_cast<u32x4*>(dst + 32) = w;
    }
    // transposed layout, per 32-key block: chunk 2 h + s = hi(y[16 s + 8 h ..+7]), chunk 4 + 2 h + s = lo(same),
    // y[q] = X[perm32(q)]; one thread per (row, block, h, s)
    for (int i = tid; i < 2 * heads * 32 * (PK / 32) * 4; i += 256) {
      const int hs = i & 3, blk = (i >> 2) % (PK / 32), ch = (i / (4 * (PK / 32))) % 32, hk = i / (4 * (PK / 32) * 32);
      const int kind = hk / heads, head = hk - kind * heads;
      E* dbase = kind ? Vt : Kt;
      if (!dbase) continue;
      const int h = hs >> 1, sst = hs & 1;
      unsigned short hb[8], lb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        split_bf16(tile[(hk * PK + blk * 32 + perm32(16 * sst + 8 * h + e)) * 32 + ch], hb[e], lb[e]);
      char* dst = reinterpret_cast<char*>(dbase + (((size_t)b * all_heads + head0 + head) * 32 + ch) * Np + n0 + blk * 32) +
                  (2 * h + sst) * 16;
      u32x4 w;
      __builtin_memcpy(&w, hb, 16);
      *reinterpret_cast<u32x4*>(dst) = w;
      __builtin_memcpy(&w, lb, 16);
      *reinterpret_cast<u32x4*>(dst + 64) = w;
    }
    return;
  }
  constexpr int EPC = 16 / sizeof(E);      // elements per 16-byte chunk
  constexpr int CPR = 32 / EPC;            // chunks per 32-element row
  // row layout: (head, key) rows of 32 elements, 64 keys contiguous per head
  for (int i = tid; i < 2 * heads * PK * CPR; i += 256) {
    const int chunk = i % CPR, key = (i / CPR) % PK, hk = i / (CPR * PK);   // hk = kind * heads + head
    const int kind = hk / heads, head = hk - kind * heads;
    E* dst = (kind ? Vr : Kr) + (((size_t)b * all_heads + head0 + head) * Np + n0 + key) * 32 + chunk * EPC;
    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(tile + (hk * PK + key) * 32 + chunk * EPC);
  }
  // transposed layout: (head, channel) rows of Np keys; this tile's 64 keys are contiguous, permuted inside each 32
  constexpr int CPT = PK / EPC;            // chunks per (head, channel) row segment of the tile
  for (int i = tid; i < 2 * heads * 32 * CPT; i += 256) {
    const int chunk = i % CPT, ch = (i / CPT) % 32, hk = i / (CPT * 32);
    const int kind = hk / heads, head = hk - kind * heads;
    E* dbase = kind ? Vt : Kt;
    if (!dbase) continue;
    E tmp[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int l = chunk * EPC + e;                       // position inside the tile
      const int keyp = (l & ~31) + perm32(l & 31);         // Xt[.., ch, l] = X[.., perm(l), ch]
      tmp[e] = tile[(hk * PK + keyp) * 32 + ch];
    }
    E* dst = dbase + (((size_t)b * all_heads + head0 + head) * 32 + ch) * Np + n0 + chunk * EPC;
    u32x4 w;
    __builtin_memcpy(&w, tmp, 16);
    *reinterpret_cast<u32x4*>(dst) = w;
  }
}

// dK, dV [B'][h][Np][32] float (row layout) -> dk, dv rows (B', N, C) with row stride ld
__global__ __launch_bounds__(256) void unpack_dkv_kernel(const float* __restrict__ dK, const float* __restrict__ dV,
                                                         float* __restrict__ dk, float* __restrict__ dv, long long ld,
                                                         long long pstride, int N, int Np, int heads, int c) {
  const int b = blockIdx.y, n0 = blockIdx.x * PK, tid = threadIdx.x;
  const int C = heads * c;
  for (int i = tid; i < PK * C; i += 256) {
    const int key = i / C, ch = i - key * C;
    if (n0 + key >= N) continue;
    const size_t s = (((size_t)b * heads + ch / c) * Np + n0 + key) * 32 + ch % c;
    const size_t o = ((size_t)b * pstride + n0 + key) * ld + ch;
    dk[o] = dK[s];
    dv[o] = dV[s];
  }
}

}  // namespace

extern "C" int bevr_pack_kv(const float* k, const float* v, long long ld, long long pstride, int n_prob, int N, int Np,
                            int heads, int c, int precision, void* Kr, void* Vr, void* Kt, void* Vt, void* stream) {
  if (!k || !v || !Kr || !Vr) return BEVR_E_NULL;
  if (n_prob <= 0 || N <= 0 || Np < N || Np % PK || heads <= 0 || c <= 0 || c > 32 || ld < (long long)heads * c ||
      pstride < N)
    return BEVR_E_SHAPE;
  if (precision < BEVR_PREC_F32 || precision > BEVR_PREC_BF16X3) return BEVR_E_PRECISION;
  if (!bevr_aligned16(Kr) || !bevr_aligned16(Vr) || (Kt && !bevr_aligned16(Kt)) || (Vt && !bevr_aligned16(Vt)))
    return BEVR_E_ALIGN;
  const size_t eb = is16(precision) ? 2 : 4;
  const int hg = (int)(32768 / (2 * PK * 32 * eb));
  const dim3 grid(Np / PK, n_prob, (heads + hg - 1) / hg);
  const size_t lds = (size_t)2 * (heads < hg ? heads : hg) * PK * 32 * eb;
  hipStream_t st = (hipStream_t)stream;
  if (precision == BEVR_PREC_F16)
    hipLaunchKernelGGL(pack_kv_kernel<half_bits>, grid, dim3(256), lds, st, k, v, ld, pstride, N, Np, heads, c,
                       (half_bits*)Kr, (half_bits*)Vr, (half_bits*)Kt, (half_bits*)Vt);
  else if (precision == BEVR_PREC_BF16)
    hipLaunchKernelGGL(pack_kv_kernel<unsigned short>, grid, dim3(256), lds, st, k, v, ld, pstride, N, Np, heads, c,
                       (unsigned short*)Kr, (unsigned short*)Vr, (unsigned short*)Kt, (unsigned short*)Vt);
  else if (precision == BEVR_PREC_BF16X3)
    hipLaunchKernelGGL((pack_kv_kernel<float, true>), grid, dim3(256), lds, st, k, v, ld, pstride, N, Np, heads, c,
                       (float*)Kr, (float*)Vr, (float*)Kt, (float*)Vt);
  else
    hipLaunchKernelGGL(pack_kv_kernel<float>, grid, dim3(256), lds, st, k, v, ld, pstride, N, Np, heads, c, (float*)Kr,
                       (float*)Vr, (float*)Kt, (float*)Vt);
  return (int)hipGetLastError();
}

extern "C" int bevr_unpack_dkv(const float* dK, const float* dV, float* dk, float* dv, long long ld, long long pstride,
                               int n_prob, int N, int Np, int heads, int c, void* stream) {
  if (!dK || !dV || !dk || !dv) return BEVR_E_NULL;
  if (n_prob <= 0 || N <= 0 || Np < N || Np % PK || heads <= 0 || c <= 0 || c > 32 || ld < (long long)heads * c ||
      pstride < N)
    return BEVR_E_SHAPE;
  hipLaunchKernelGGL(unpack_dkv_kernel, dim3(Np / PK, n_prob), dim3(256), 0, (hipStream_t)stream, dK, dV, dk, dv, ld,
                     pstride, N, Np, heads, c);
  return (int)hipGetLastError();
}
