// Query-side backward of the tap kernels (attn_tap.h).  With P[n][q] = exp2(S[n][q] - LSE[q]) recomputed as in the forward,
//     dP[n][q] = dO_q . V_n = sum_t w_t(n) H[t][q] + Hb[q],         H[t][q] = Vpix_t . dO_q, Hb = bv . dO_q
//     dS[n][q] = ln2 P (dP - delta[q])                              (the caller folds ln2 and delta into H and Hc)
//     dG[k][q] = sum_n A[n][k] dS[n][q],    A = [w | Wc]:   rows 0..15 the gradient of G (row TAP_ONE: of Gb), rows 16..31
//                                                           the gradient of the chunk's 16 table cells shifted by q
// (reference: the backward of model/SCA_deform_attn.py:331-413 through autograd).  dQ and dKpix follow from dG by two thin
// GEMMs in the caller; dV needs nothing from here (dVpix = Rn^T dO with the forward's R).
//
// Same decomposition and key stream as attn_tap_fwd.hip (workgroup = one BEV column, row-block waves + the producer of
// attn_tap.h).  Per 32-key tile and 16-row block: S^T and dP^T (2 + 2 MFMAs, the SAME A operand), 8 exponentials, 8
// products, one conversion of dS to 16 bits that feeds both dG products (A = w^T and Wc^T through ds_read_b64_tr_b16
// from the images the producer wrote).  The cell half of dG lives in 4 registers per row block while the chunk origin
// stays and is flushed into the table gradient with float atomics when it changes (~1 tile in 30).
#include "attn_tap.h"

namespace {

template <int PREC, int NB>
__global__ __launch_bounds__(512, 4) void attn_tap_bwd_q_kernel(
    bevr_attn_desc d, const char* __restrict__ G, const char* __restrict__ H, const char* __restrict__ tap_ws, const char* __restrict__ table_pair,
    float* __restrict__ dG, float* __restrict__ dtable) {
  typedef LdsT L;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / d.S) * 8 + xcd;
  if (ph >= n_ph) return;
  const int j = slot % d.S;
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

  const int blk0 = wave * NB;
  bf16x8 bop[NB];     // lanes 0..31 G[q][8 kg ..], lanes 32..63 the chunk's table side
  bf16x8 hop[NB];     // lanes 0..31 H[q][8 kg ..], lanes 32..63 zero
  f32x4 ytap[NB], ycell[NB];
  size_t mqv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int blk = min(blk0 + nb, nblk - 1);
    const size_t mq = (size_t)ph * Mp + (size_t)j * d.Sp + blk * QB + li;
    mqv[nb] = mq;
    u32x4 g = {0u, 0u, 0u, 0u}, hh = {0u, 0u, 0u, 0u};
    if (kg < 2) {
      g = *reinterpret_cast<const u32x4*>(G + (mq * TAP_SLOTS + 8 * kg) * 2);
      hh = *reinterpret_cast<const u32x4*>(H + (mq * TAP_SLOTS + 8 * kg) * 2);
    }
    bop[nb] = __builtin_bit_cast(bf16x8, g);
    hop[nb] = __builtin_bit_cast(bf16x8, hh);
    ytap[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    ycell[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int a_off = (kg < 2 ? L::OFF_TAPS : L::OFF_CELLS) + li * 32 + (kg & 1) * 16;     // + tile * 1024 + sub * 512
  const int t_off = (4 * kg + (li >> 2)) * 32 + (lane & 3) * 8;                           // transposed reads: + image, + tile * 1024
  const int i_off = li * 32 + (kg & 1) * 16;
  int have = 0, org_x = 0, org_a = 0;   // the chunk in bop / ycell: allocation number (0: none yet) and origin

  // table gradient of the chunk in ycell: cell (c = kg, r) of BEV row q is table entry (org_x + c, org_a + q + r)
  float* dth = dtable + (size_t)hd * d.Wp * (d.Hp + 1);
  // Rows q and q + 1 of a block overlap in three of their four cells: cell r of row q is table row org_a + q + r.  The four
  // contributions to one table row are summed across the block's lanes first (one atomic per lane instead of four; the
  // last three lanes add the rows past the block's sixteenth on their own): 2.9x fewer atomics
  auto flush = [&]() {
    const int xc = org_x + kg + d.x_off;
    const bool col_ok = xc >= 0 && xc < d.Wp;
    float* col = dth + (size_t)(col_ok ? xc : 0) * (d.Hp + 1);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (NB > 1 && blk0 + nb >= nblk) continue;       // uniform
      const f32x4 c = ycell[nb];
      float s = c[0];
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        const float up = __shfl_up(c[r], r, QB);        // cell r of the row r lanes below
        s += li >= r ? up : 0.f;
      }
      const int y0 = org_a + (blk0 + nb) * QB + li + d.y_off;
      if (col_ok) {
        if (y0 >= 0 && y0 <= d.Hp) atomicAdd(col + y0, s);
#pragma unroll
        for (int r = 1; r < 4; ++r) {                   // table rows past the block's sixteenth
          const int y = y0 + r;
          if (li + r >= QB && y >= 0 && y <= d.Hp) atomicAdd(col + y, c[r]);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) ycell[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  auto tile = [&](const char* base, int t) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + a_off + t * 1024));
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + a_off + t * 1024 + 512));
    const bf16x8 wt = lds_tr8(base + L::OFF_TAPS + t_off + t * 1024, 512);
    const bf16x8 wct = lds_tr8(base + L::OFF_CELLS + t_off + t * 1024, 512);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (NB > 1 && blk0 + nb >= nblk) continue;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 s0 = mfma16<PREC>(a0, bop[nb], z4);
      const f32x4 s1 = mfma16<PREC>(a1, bop[nb], z4);
      const f32x4 p0 = mfma16<PREC>(a0, hop[nb], z4);
      const f32x4 p1 = mfma16<PREC>(a1, hop[nb], z4);
      u32x4 dsw;
      dsw[0] = Half<PREC>::pack2(fast_exp2(s0[0]) * p0[0], fast_exp2(s0[1]) * p0[1]);
      dsw[1] = Half<PREC>::pack2(fast_exp2(s0[2]) * p0[2], fast_exp2(s0[3]) * p0[3]);
      dsw[2] = Half<PREC>::pack2(fast_exp2(s1[0]) * p1[0], fast_exp2(s1[1]) * p1[1]);
      dsw[3] = Half<PREC>::pack2(fast_exp2(s1[2]) * p1[2], fast_exp2(s1[3]) * p1[3]);
      const bf16x8 ds8 = __builtin_bit_cast(bf16x8, dsw);
      ytap[nb] = mfma16<PREC>(wt, ds8, ytap[nb]);
      ycell[nb] = mfma16<PREC>(wct, ds8, ycell[nb]);
    }
  };

  for (int e = 0;; ++e) {
    __syncthreads();
    const char* base = smem + (e & 1) * L::BUF;
    const u32x4 ct = *reinterpret_cast<const u32x4*>(base + L::OFF_CT);
    const u32x4 og = *reinterpret_cast<const u32x4*>(base + L::OFF_ORG);
    const int fl = __builtin_amdgcn_readfirstlane((int)ct[0]);
    if (fl & 4) break;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int al = __builtin_amdgcn_readfirstlane((int)ct[1 + t]);
      if (al != have) {   // uniform, rare: another chunk origin -- hand the old chunk's table gradient over first
        if (have != 0) flush();
        have = al;
        org_x = __builtin_amdgcn_readfirstlane((int)og[2 * t]);
        org_a = __builtin_amdgcn_readfirstlane((int)og[2 * t + 1]);
        if (kg >= 2) {
          const char* img = ring + (al & (L::RING - 1)) * img_bytes + i_off;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            bop[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + min(blk0 + nb, nblk - 1) * 512));
        }
      }
      tile(base, t);
    }
  }
  if (have != 0) flush();

#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (blk0 + nb >= nblk) continue;
    *reinterpret_cast<f32x4*>(dG + mqv[nb] * TAP_SLOTS + 4 * kg) = ytap[nb];
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* G, const void* H, const void* tap_ws,
           const float* table_pair, float* dG, float* dtable, hipStream_t st) {
  typedef LdsT L;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * d.S;
  const int nblk = (d.S + QB - 1) / QB;
  const size_t lds = 2 * L::BUF + (size_t)L::RING * nblk * QB * 32;
  if (lds > 160 * 1024 || nblk > 28) return BEVR_E_SHAPE;
  const int nb = nblk <= 7 ? 1 : nblk <= 14 ? 2 : 4;
  const int n_cw = (nblk + nb - 1) / nb;
  const dim3 block(64 * (n_cw + 1));
#define BEVR_TAP_LAUNCH(NB_)                                                                                            \
  hipLaunchKernelGGL((attn_tap_bwd_q_kernel<PREC, NB_>), dim3(grid), block, lds, st, d, (const char*)G, (const char*)H,     \
                     (const char*)tap_ws, (const char*)table_pair, dG, dtable)
  if (nb == 1) BEVR_TAP_LAUNCH(1);
  else if (nb == 2) BEVR_TAP_LAUNCH(2);
  else BEVR_TAP_LAUNCH(4);
#undef BEVR_TAP_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_tap_bwd_q(const bevr_attn_desc* d, const void* G, const void* H, const void* tap_ws,
                                   const float* table_pair, float* dG, float* dtable, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!G || !H || !tap_ws || !table_pair || !dG || !dtable) return BEVR_E_NULL;
  if (d->groups != 1) return BEVR_E_SHAPE;
  if (!bevr_aligned16(G) || !bevr_aligned16(H) || !bevr_aligned16(tap_ws) || !bevr_aligned16(table_pair) || !bevr_aligned16(dG))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16) return launch<BEVR_PREC_BF16>(*d, G, H, tap_ws, table_pair, dG, dtable, st);
  if (d->precision == BEVR_PREC_F16) return launch<BEVR_PREC_F16>(*d, G, H, tap_ws, table_pair, dG, dtable, st);
  return BEVR_E_PRECISION;
}
