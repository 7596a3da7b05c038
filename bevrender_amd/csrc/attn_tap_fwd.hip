// Forward of the tap kernels (attn_tap.h): R[slot][q] = sum_n w_slot(n) exp2(S[n][q] - mref[q]) over a key segment whose
// keys sample inside the top-left TAP_R x TAP_C feature pixels,
//     S[n][q] = sum_t w_t(n) G[t][q] + Gb[q] + bias[n][q]          (model/SCA_deform_attn.py:331-402 of the reference),
// row TAP_ONE of R = the softmax denominator.  No K / V operand exists: the caller forms G = scale Kpix Q^T before and
// O = R Vpix after the launch (two thin GEMMs) and merges the result with the other key segment of the same softmax.
//
// Work split: workgroup = ONE BEV column j of one (problem, head); wave w owns the 16-row blocks [w NB, (w + 1) NB) of the
// column, the LAST wave is the PRODUCER.  Per emission (two 32-key tiles) the producer turns the keys' records into the A
// operand ([12 tap weights, 0, 0, dead, 1 | 16 bias-cell weights] per key, 64 B) in LDS, and, when a tile's chunk origin
// differs from the previous one, the table side Tsh[cell][row] of that chunk for every BEV row of the column into a ring of
// four LDS images: the row-block waves never touch global memory inside the loop.  One barrier per emission.
//
// ANY key set is handled: a tile whose taps do not fit one 4 x 4 chunk for this column is emitted several times, each
// time with the keys of one chunk live and the others masked (at least one key per pass); a cell-sorted segment
// (ops.cell_order) needs that for ~0.1 % of its tiles.
//
// Softmax reference: STATIC.  mref[q] is given by the caller (an upper bound of the row's logits minus a headroom, so
// that no weight can overflow); the loop carries no running maximum, no rescale and no row sum.  A row whose weights
// all underflowed against that reference (bound looser than ~190 binades: exploding activations) is flagged per column
// and recomputed by the EXACT instantiation (online maximum), which overwrites R and mref of the flagged columns.
#include "attn_tap.h"

namespace {

// NB: 16-row blocks per row-block wave (at most 7 row-block waves + the producer: fewer, fatter waves beat one wave per
// block by 1.5x -- the barrier is cheaper and the A operand is read once per wave).  EXACT: online maximum (the recompute
// pass of flagged columns).
template <int PREC, int NB, bool EXACT>
__global__ __launch_bounds__(512, (NB == 4 ? 4 : 6)) void attn_tap_fwd_kernel(
    bevr_attn_desc d, const char* __restrict__ G, const char* __restrict__ tap_ws,
    const char* __restrict__ table_pair, float* __restrict__ mref, float* __restrict__ R, int* __restrict__ flags) {
  typedef LdsT L;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / d.S) * 8 + xcd;
  if (ph >= n_ph) return;
  const int j = slot % d.S;
  if constexpr (EXACT) {
    if (flags[ph * d.S + j] == 0) return;
  }
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int tid = threadIdx.x, n_wave = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, kg = lane >> 4;
  const int Mp = d.S * d.Sp;
  const int nblk = (d.S + QB - 1) / QB;
  const int rows_img = nblk * QB;
  const int img_bytes = rows_img * 32;
  char* ring = smem + 2 * L::BUF;
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const float jrx = (float)j * rx;

  if (wave == n_wave - 1) {
    const TapRec* recs = reinterpret_cast<const TapRec*>(tap_ws) + (size_t)prob * d.Np;
    const StepBox* box = reinterpret_cast<const StepBox*>(tap_ws + tap_ws_box_offset(d)) + (size_t)prob * (d.Np / 32);
    tap_producer<PREC>(d, smem, ring, img_bytes, rows_img, recs, box, tbl, jrx, lane);
    return;
  }

  // ---- row-block waves ------------------------------------------------------------------------------------
  const int blk0 = wave * NB;
  bf16x8 bop[NB];     // B operand: lanes 0..31 G[q][8 kg ..] (constant), lanes 32..63 the chunk's table side (per origin)
  f32x4 r[NB];        // R[slot 4 kg + e][q]
  float sh[NB];       // EXACT: the running maximum relative to mref
  size_t mqv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int blk = min(blk0 + nb, nblk - 1);
    const size_t mq = (size_t)ph * Mp + (size_t)j * d.Sp + blk * QB + li;
    mqv[nb] = mq;
    u32x4 g = {0u, 0u, 0u, 0u};
    if (kg < 2) g = *reinterpret_cast<const u32x4*>(G + (mq * TAP_SLOTS + 8 * kg) * 2);
    bop[nb] = __builtin_bit_cast(bf16x8, g);
    r[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    sh[nb] = -3.0e38f;
  }
  // this lane's LDS addresses inside a buffer
  const int a_off = (kg < 2 ? L::OFF_TAPS : L::OFF_CELLS) + li * 32 + (kg & 1) * 16;     // + tile * 1024 + sub * 512
  const int t_off = L::OFF_TAPS + (4 * kg + (li >> 2)) * 32 + (lane & 3) * 8;             // + tile * 1024, second block + 512
  const int i_off = li * 32 + (kg & 1) * 16;                                             // in a table image, + block * 512
  // the B operand exists once per tile slot of an emission (their chunk origins may differ); lanes 0..31 of both hold G
  bf16x8 bop1[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bop1[nb] = bop[nb];
  int have0 = 0, have1 = 0;      // allocation numbers of the table images in bop / bop1 (0: the zeroed image)

  // one 32-key tile against one row block: S^T (two 16-key sub-tiles) -> weights -> R += w^T P
  auto tile = [&](const bf16x8& a0, const bf16x8& a1, const bf16x8& wt, const bf16x8& b, int nb) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s0 = mfma16<PREC>(a0, b, z4);
    f32x4 s1 = mfma16<PREC>(a1, b, z4);
    if constexpr (EXACT) {
      float tm = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      tm = fmaxf(tm, __shfl_xor(tm, 16));
      tm = fmaxf(tm, __shfl_xor(tm, 32));
      const float mn = fmaxf(sh[nb], tm);
      const float al2 = fast_exp2(sh[nb] - mn);
      r[nb] *= al2;
      sh[nb] = mn;
      s0 -= mn;
      s1 -= mn;
    }
    u32x4 pw;
    pw[0] = Half<PREC>::pack2(fast_exp2(s0[0]), fast_exp2(s0[1]));
    pw[1] = Half<PREC>::pack2(fast_exp2(s0[2]), fast_exp2(s0[3]));
    pw[2] = Half<PREC>::pack2(fast_exp2(s1[0]), fast_exp2(s1[1]));
    pw[3] = Half<PREC>::pack2(fast_exp2(s1[2]), fast_exp2(s1[3]));
    r[nb] = mfma16<PREC>(wt, __builtin_bit_cast(bf16x8, pw), r[nb]);
  };

  for (int e = 0;; ++e) {
    __syncthreads();
    const char* base = smem + (e & 1) * L::BUF;
    const u32x4 ct = *reinterpret_cast<const u32x4*>(base + L::OFF_CT);
    const int fl = __builtin_amdgcn_readfirstlane((int)ct[0]);
    if (fl & 4) break;
    const int al0 = __builtin_amdgcn_readfirstlane((int)ct[1]), al1 = __builtin_amdgcn_readfirstlane((int)ct[2]);
    if (al0 != have0) {   // uniform, rare: another chunk origin
      have0 = al0;
      if (kg >= 2) {
        const char* img = ring + (al0 & (L::RING - 1)) * img_bytes + i_off;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          bop[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + min(blk0 + nb, nblk - 1) * 512));
      }
    }
    if (al1 != have1) {
      have1 = al1;
      if (kg >= 2) {
        const char* img = ring + (al1 & (L::RING - 1)) * img_bytes + i_off;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          bop1[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + min(blk0 + nb, nblk - 1) * 512));
      }
    }
    // both tile slots, unconditionally: a slot without live keys holds masked keys only (weight 0)
    const bf16x8 a00 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + a_off));
    const bf16x8 a01 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + a_off + 512));
    const bf16x8 a10 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + a_off + 1024));
    const bf16x8 a11 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + a_off + 1536));
    const bf16x8 wt0 = lds_tr8(base + t_off, 512);
    const bf16x8 wt1 = lds_tr8(base + t_off + 1024, 512);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (NB > 1 && blk0 + nb >= nblk) continue;
      tile(a00, a01, wt0, bop[nb], nb);
      tile(a10, a11, wt1, bop1[nb], nb);
    }
  }

  // ---- epilogue: R[q][4 kg .. 4 kg + 3]; flag the column if a row's mass is not a healthy number -----------------
  bool bad = false;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (blk0 + nb >= nblk) continue;
    *reinterpret_cast<f32x4*>(R + mqv[nb] * TAP_SLOTS + 4 * kg) = r[nb];
    const int row = (blk0 + nb) * QB + li;
    if constexpr (EXACT) {
      if (kg == 0) mref[mqv[nb]] += sh[nb];
    } else {
      const float l = r[nb][3];
      if (kg == 3 && row < d.S && !(l >= 7.9e-31f && l < 3.0e38f)) bad = true;
    }
  }
  if constexpr (!EXACT) {
    if (__any(bad) && lane == 0) flags[ph * d.S + j] = 1;
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* G, const void* tap_ws, const float* table_pair,
           float* mref, float* R, int* flags, hipStream_t st) {
  typedef LdsT L;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * d.S;
  const int nblk = (d.S + QB - 1) / QB;
  const size_t lds = 2 * L::BUF + (size_t)L::RING * nblk * QB * 32;
  if (lds > 160 * 1024) return BEVR_E_SHAPE;
  const int nb = nblk <= 7 ? 1 : nblk <= 14 ? 2 : 4;      // row blocks per wave: at most 7 row-block waves + the producer
  if (nblk > 28) return BEVR_E_SHAPE;
  const int n_cw = (nblk + nb - 1) / nb;
  const dim3 block(64 * (n_cw + 1));
#define BEVR_TAP_LAUNCH(NB_, EX_)                                                                                     \
  hipLaunchKernelGGL((attn_tap_fwd_kernel<PREC, NB_, EX_>), dim3(grid), block, lds, st, d, (const char*)G,          \
                     (const char*)tap_ws, (const char*)table_pair, mref, R, flags)
  for (int ex = 0; ex < 2; ++ex) {
    if (nb == 1) { if (ex) BEVR_TAP_LAUNCH(1, true); else BEVR_TAP_LAUNCH(1, false); }
    else if (nb == 2) { if (ex) BEVR_TAP_LAUNCH(2, true); else BEVR_TAP_LAUNCH(2, false); }
    else { if (ex) BEVR_TAP_LAUNCH(4, true); else BEVR_TAP_LAUNCH(4, false); }
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
  }
#undef BEVR_TAP_LAUNCH
  return BEVR_OK;
}

}  // namespace

extern "C" int bevr_attn_tap_fwd(const bevr_attn_desc* d, const void* G, const void* tap_ws,
                                 const float* table_pair, float* mref, float* R, int* flags, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!G || !tap_ws || !table_pair || !mref || !R || !flags) return BEVR_E_NULL;
  if (d->groups != 1) return BEVR_E_SHAPE;
  if (!bevr_aligned16(G) || !bevr_aligned16(tap_ws) || !bevr_aligned16(table_pair) || !bevr_aligned16(R)) return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16) return launch<BEVR_PREC_BF16>(*d, G, tap_ws, table_pair, mref, R, flags, st);
  if (d->precision == BEVR_PREC_F16) return launch<BEVR_PREC_F16>(*d, G, tap_ws, table_pair, mref, R, flags, st);
  return BEVR_E_PRECISION;
}
