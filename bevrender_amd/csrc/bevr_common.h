// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bevrender_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define BEVR_LOG2E 1.4426950408889634f
#define BEVR_LN2 0.6931471805599453f
#define BEVR_NEG_BIG (-1.0e30f)

// Row of the 32x32 MFMA accumulator held in register r by lane-half hi (cdna guide section 3):
//   row = (r & 3) + 8 * (r >> 2) + 4 * hi ; column = lane & 31.
__device__ __forceinline__ constexpr int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// In-32 permutation used for every operand that is contracted against an accumulator tile's rows:
// index with bits 2 and 3 swapped (an involution).
__host__ __device__ __forceinline__ constexpr int perm32(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf without branches (Abramowitz & Stegun 7.1.26: |error| < 1.5e-7, float rounding of the same size): the library erff
// branches on |x| and a wave runs both sides (offset heads: 5 GELUs per lane and pixel were a third of the kernel).
// Returns erf(x) and, through e, exp(-x^2) (the gradient's Gaussian, for free).
__device__ __forceinline__ float erf_as(float x, float& e) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e = __expf(-ax * ax);
  return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float y) {
  float e;
  return 0.5f * y * (1.0f + erf_as(y * 0.70710678118654752f, e));
}
__device__ __forceinline__ float gelu_erf_grad(float y) {
  float e;      // exp(-y^2 / 2)
  const float er = erf_as(y * 0.70710678118654752f, e);
  return 0.5f * (1.0f + er) + y * 0.3989422804014327f * e;
}

// round-to-nearest-even f32 -> bf16 pair packed in one dword (plain casts lower to v_cvt_pk_bf16_f32).
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

template <int PREC> struct Elem;
template <> struct Elem<BEVR_PREC_F32> { typedef float type; static constexpr int bytes = 4; };
template <> struct Elem<BEVR_PREC_BF16> { typedef __bf16 type; static constexpr int bytes = 2; };
template <> struct Elem<BEVR_PREC_F16> { typedef _Float16 type; static constexpr int bytes = 2; };
template <> struct Elem<BEVR_PREC_BF16X3> { typedef float type; static constexpr int bytes = 4; };   // f32 storage

// the two 16-bit operand modes share every layout and all storage (kept as raw bits in bf16x8 / uint32 registers); what
// differs is the arithmetic on the bits
__host__ __device__ constexpr bool is16(int prec) { return prec == BEVR_PREC_BF16 || prec == BEVR_PREC_F16; }
template <int PREC> struct Half;
template <> struct Half<BEVR_PREC_BF16> {
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
  static __device__ __forceinline__ float lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
  static __device__ __forceinline__ float hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
  }
  static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  static constexpr float NEG_BIG = -1.0e30f;   // "masked" entry of a 16-bit table window
  static constexpr float SHIFT = 0.f;          // binades between the forward's softmax reference and the running maximum
};
template <> struct Half<BEVR_PREC_F16> {
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));   // v_cvt_pk_f16_f32, nearest even
  }
  static __device__ __forceinline__ float lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
  static __device__ __forceinline__ float hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
  }
  static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
  // fp16 cannot hold -1e30 (it would become -inf, and -inf times a zero tap weight is NaN): any value whose exp2 is 0
  static constexpr float NEG_BIG = -60000.0f;
  // fp16's normal range ends at 2^-14: the forward keeps its reference 10 binades under the running maximum, so that
  // softmax weights down to 2^-24 of the largest are normal numbers (they may reach 2^12; fp16 holds 2^15)
  static constexpr float SHIFT = 10.f;
};

// A or B operand fragment of one 32-wide tile with the 32-deep contraction held by this lane:
//   bf16: 2 k-steps x 8 elements  (element j of step s <-> contraction index 16 s + 8 hi + j)
//   f32 : 16 k-steps x 1 element  (step s <-> contraction index 16 hi + s)
// In both cases the lane reads 16 *consecutive-in-memory* elements pairs: bf16 -> two 16-B chunks at
// element offsets 8 hi and 16 + 8 hi; f32 -> 16 floats at element offset 16 hi.
template <int PREC> struct Frag;
template <> struct Frag<BEVR_PREC_BF16> {
  bf16x8 v[2];
  // row: pointer to the 32 contraction-contiguous elements of this lane's row
  __device__ __forceinline__ void load(const void* row, int hi) {
    const u32x4* p = reinterpret_cast<const u32x4*>(row);
    u32x4 a = p[hi], b = p[2 + hi];
    v[0] = __builtin_bit_cast(bf16x8, a);
    v[1] = __builtin_bit_cast(bf16x8, b);
  }
};
template <> struct Frag<BEVR_PREC_F16> : Frag<BEVR_PREC_BF16> {};   // same bits, same loads
template <> struct Frag<BEVR_PREC_F32> {
  float v[16];
  __device__ __forceinline__ void load(const void* row, int hi) {
    const f32x4* p = reinterpret_cast<const f32x4*>(row) + 4 * hi;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 t = p[k];
      v[4 * k + 0] = t[0]; v[4 * k + 1] = t[1]; v[4 * k + 2] = t[2]; v[4 * k + 3] = t[3];
    }
  }
};

// BEVR_PREC_BF16X3: the f32-tolerance mode on the bf16 matrix cores.  Every matrix operand is SPLIT x = hi + lo
// (hi = bf16(x), lo = bf16(x - hi), both round-to-nearest) and a product is three MFMAs,
//   a b ~= a_lo b_hi + a_hi b_lo + a_hi b_hi,   dropped: a_lo b_lo <= 2^-16 |a b|   (f32 accumulation as always):
// three 32x32x16 bf16 MFMAs (3 x 32 clk) replace eight 32x32x2 f32 MFMAs (8 x 64 clk) per 16 contraction indices.
// The packed operands in HBM and LDS keep the f32 modes' sizes, strides and addressing -- a 32-element row or 32-block is
// 128 bytes, a lane reads the same 64 of them -- but hold the two bf16 planes instead of floats (formats below: the
// caller / bevr_pack_kv split ONCE per element; the kernels split only what they produce themselves, P and dS).
// Everything per-pair (bias taps, table window, softmax) is the f32 modes' code.  The fragment types keep the f32
// containers as raw bits, so that every copy (LDS slots, staging) is shared with BEVR_PREC_F32:
//   Frag<X3>.v[0..3] = hi plane of k-step 0 (8 bf16), v[4..7] = hi of k-step 1, v[8..11] / v[12..15] = the lo planes.
//   row layout, per 16-element half [e0..e15] (64 B):  hi(e0..e7) | hi(e8..e15) | lo(e0..e7) | lo(e8..e15)
//   transposed (perm32) layout, per 32-block (128 B), chunk c = 16 B: with y[q] the block in the f32 mode's order,
//     chunk 2 h + s = hi(y[16 s + 8 h .. + 7]),  chunk 4 + 2 h + s = lo(same)        (h = lane half, s = k-step)
template <> struct Frag<BEVR_PREC_BF16X3> : Frag<BEVR_PREC_F32> {};   // same loads (load_perm binds to the base)
struct Split8 { bf16x8 hi, lo; };
__device__ __forceinline__ Split8 split8(const float* x) {
  u32x4 h, l;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = pack_bf16x2(x[2 * k], x[2 * k + 1]);
    const float f0 = __builtin_bit_cast(float, h[k] << 16), f1 = __builtin_bit_cast(float, h[k] & 0xffff0000u);
    l[k] = pack_bf16x2(x[2 * k] - f0, x[2 * k + 1] - f1);
  }
  return Split8{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l)};
}
// the planes of 4 raw dwords each out of a fragment container
__device__ __forceinline__ bf16x8 raw8(const float* v) {
  return __builtin_bit_cast(bf16x8, f32x4{v[0], v[1], v[2], v[3]});
}
__device__ __forceinline__ void put8(float* v, bf16x8 b) {
  const f32x4 t = __builtin_bit_cast(f32x4, b);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ f32x16 mma_split(const Split8& a, const Split8& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);   // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
  return acc;
}

// acc += A(rows x 32) * B(32 x cols) for one 32x32 tile, both operands as Frag (contraction over the
// fragment's 32 elements in matching order).
__device__ __forceinline__ f32x16 mma_frag(const Frag<BEVR_PREC_BF16>& a, const Frag<BEVR_PREC_BF16>& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0], b.v[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1], b.v[1], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 mma_frag(const Frag<BEVR_PREC_F16>& a, const Frag<BEVR_PREC_F16>& b, f32x16 acc) {
  acc = Half<BEVR_PREC_F16>::mfma(a.v[0], b.v[0], acc);
  acc = Half<BEVR_PREC_F16>::mfma(a.v[1], b.v[1], acc);
  return acc;
}
__device__ __forceinline__ f32x16 mma_frag(const Frag<BEVR_PREC_F32>& a, const Frag<BEVR_PREC_F32>& b, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ f32x16 mma_frag(const Frag<BEVR_PREC_BF16X3>& a, const Frag<BEVR_PREC_BF16X3>& b, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
    acc = mma_split(Split8{raw8(a.v + 4 * s), raw8(a.v + 8 + 4 * s)}, Split8{raw8(b.v + 4 * s), raw8(b.v + 8 + 4 * s)}, acc);
  return acc;
}

// acc += A * X where X is an ACCUMULATOR-layout tile (rows = contraction index, cols = lanes) used as
// the B operand.  The A fragment must have been loaded from memory whose in-32 order is perm32 (see
// bevrender_hip.h, Vt): then element j of step s of the bf16 fragment is contraction row
// 16 s + 8 (j >> 2) + 4 hi + (j & 3) = crow(8 s + j, hi), and f32 step t is crow(t, hi).
__device__ __forceinline__ f32x16 mma_acc_b(const Frag<BEVR_PREC_BF16>& a, const f32x16& x, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = pack_bf16x2(x[8 * s + 2 * k], x[8 * s + 2 * k + 1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[s], __builtin_bit_cast(bf16x8, w), acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ f32x16 mma_acc_b(const Frag<BEVR_PREC_F16>& a, const f32x16& x, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = Half<BEVR_PREC_F16>::pack2(x[8 * s + 2 * k], x[8 * s + 2 * k + 1]);
    acc = Half<BEVR_PREC_F16>::mfma(a.v[s], __builtin_bit_cast(bf16x8, w), acc);
  }
  return acc;
}
__device__ __forceinline__ f32x16 mma_acc_b(const Frag<BEVR_PREC_F32>& a, const f32x16& x, f32x16 acc) {
  // memory order perm32: fragment element t (within this lane-half's 16) sits at position
  // 16 (t >> 3) + 8 hi + (t & 7) of the 32-block; Frag::load read positions 16 hi .. 16 hi + 15, which is
  // NOT that set -- the f32 A operand for an accumulator contraction is loaded with load_perm below.
#pragma unroll
  for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], x[t], acc, 0, 0, 0);
  return acc;
}

// the accumulator-layout operand is split here; element j of k-step s <-> x[8 s + j], as in the 16-bit modes
__device__ __forceinline__ f32x16 mma_acc_b(const Frag<BEVR_PREC_BF16X3>& a, const f32x16& x, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float xs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xs[k] = x[8 * s + k];
    acc = mma_split(Split8{raw8(a.v + 4 * s), raw8(a.v + 8 + 4 * s)}, split8(xs), acc);
  }
  return acc;
}

// Load an A fragment for mma_acc_b from a 32-block stored in perm32 order.
__device__ __forceinline__ void load_perm(Frag<BEVR_PREC_BF16>& f, const void* row, int hi) { f.load(row, hi); }
__device__ __forceinline__ void load_perm(Frag<BEVR_PREC_F16>& f, const void* row, int hi) { f.load(row, hi); }
__device__ __forceinline__ void load_perm(Frag<BEVR_PREC_F32>& f, const void* row, int hi) {
  // wanted: contraction row crow(t, hi) = 8 (t>>2) + 4 hi + (t&3); its perm32 position is
  // 16 (t>>3) + 8 hi + (t & 7).
  const f32x4* p = reinterpret_cast<const f32x4*>(row);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 t0 = p[4 * half + 2 * hi], t1 = p[4 * half + 2 * hi + 1];
    f.v[8 * half + 0] = t0[0]; f.v[8 * half + 1] = t0[1]; f.v[8 * half + 2] = t0[2]; f.v[8 * half + 3] = t0[3];
    f.v[8 * half + 4] = t1[0]; f.v[8 * half + 5] = t1[1]; f.v[8 * half + 6] = t1[2]; f.v[8 * half + 7] = t1[3];
  }
}

// ---------------------------------------------------------------------------------------------
// Bias-tap addressing shared by the attention kernels (global-memory gather path).
// key constants, packed per key by the kernels' prologue: Aoff (byte offset of row floor(a)+y_off
// in column x_off), fy, clamped b, 1 - fy.
struct KeyC { int aoff; float fy; float b; float wy0; };

__device__ __forceinline__ KeyC make_keyc(float a, float b, const bevr_attn_desc& d) {
  float aL = -(float)(d.Sp + 1), aU = (float)(d.Ht + 1);
  float half = (float)((d.Wt) / 2);  // ceil((Wt-1)/2)
  float bL = -(half + 2.0f), bU = (float)(d.Wt + 1);
  // NaN-safe clamps (fminf/fmaxf return the non-NaN operand)
  a = fminf(fmaxf(a, aL), aU);
  b = fminf(fmaxf(b, bL), bU);
  float af = floorf(a);
  KeyC k;
  k.aoff = (((int)af + d.y_off) + d.x_off * d.Hp) * 8;
  k.fy = a - af;
  k.wy0 = 1.0f - k.fy;
  k.b = b;
  return k;
}

// ---------------------------------------------------------------------------------------------
// Attention dropout (reference model/SCA_deform_attn.py:155,402-409, model/TSA_deform_attn.py:90,313-323: nn.Dropout on
// the softmax weights, the kept ones scaled by 1 / (1 - p)).  The keep decision of the pair (problem-head ph, packed
// query mq, key n) is a pure function of (seed, ph, mq, n): the forward and both backward kernels evaluate the same
// function instead of storing an (M x N) mask.  keep iff the hash's top 16 bits >= thr16, thr16 = round(p * 65536).
// bevrender_amd/ops.py:dropout_keep_mask is the same function on the host (the tests build the oracle's mask with it).
__host__ __device__ __forceinline__ uint32_t bevr_drop_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t bevr_drop_row(uint32_t seed, uint32_t ph, uint32_t mq) {
  return seed ^ (ph * 0x9E3779B1u) ^ (mq * 0x85EBCA77u);
}
__host__ __device__ __forceinline__ bool bevr_drop_keep(uint32_t row, uint32_t n, uint32_t thr16) {
  return (bevr_drop_mix(row ^ (n * 0xC2B2AE3Du)) >> 16) >= thr16;
}
// The dropout variants of the region kernels are separate translation units (attn_*_drop.hip: #define BEVR_DROP 1 and
// #include the kernel's source): the kernels without dropout stay byte for byte what they were.
#ifndef BEVR_DROP
#define BEVR_DROP 0
#endif
#if BEVR_DROP
#define BEVR_DROP_PARAMS , unsigned drop_thr, unsigned drop_seed
#define BEVR_DROP_ARGS , drop_thr, drop_seed
#else
#define BEVR_DROP_PARAMS
#define BEVR_DROP_ARGS
#endif

static inline int bevr_check_desc(const bevr_attn_desc* d) {
  if (!d) return BEVR_E_NULL;
  if (d->n_prob <= 0 || d->q_div <= 0 || d->n_prob % d->q_div) return BEVR_E_SHAPE;
  if (d->heads <= 0 || d->groups <= 0 || d->heads % d->groups) return BEVR_E_SHAPE;
  if (d->S < 2 || d->Sp != 32 * ((d->S + 31) / 32)) return BEVR_E_SHAPE;
  if (d->N <= 0 || d->Np < d->N || d->Np % 64) return BEVR_E_SHAPE;
  if (d->Ht != 2 * d->S - 1 || d->Wt < 1) return BEVR_E_SHAPE;
  bevr_attn_desc t = *d;
  if (bevr_attn_table_dims(&t) != 0) return BEVR_E_SHAPE;
  if (t.Hp != d->Hp || t.Wp != d->Wp || t.y_off != d->y_off || t.x_off != d->x_off) return BEVR_E_SHAPE;
  if (d->precision < BEVR_PREC_F32 || d->precision > BEVR_PREC_BF16X3) return BEVR_E_PRECISION;
  // 32-bit byte offsets into one head's pair table
  if ((long long)d->Hp * d->Wp * 8 >= (1LL << 31)) return BEVR_E_SHAPE;
  return BEVR_OK;
}

static inline int bevr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
