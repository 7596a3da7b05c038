// Attention backward, query side: dQ and the rpe-table gradient.
// Same orientation, tiling, LDS table window and XCD mapping as the forward (attn_fwd.hip): S^T[key][query]
// tiles with the query on the lane, so
//   * P^T = exp2(S^T - LSE[q]) and dS^T = ln2 * P^T * (dP^T - delta[q]) need only lane-local constants
//     (they are preloaded into the accumulators of the two MFMA chains),
//   * dQ^T[c][q] += K^T[c][key] dS^T[key][q] takes dS^T straight from the accumulator (B operand),
//   * the table gradient of a (query tile x key step) block lands inside the same bounding box the bias
//     was read from: each wave accumulates it in a private LDS window of the region's shape (plain
//     read-modify-write, 32 lanes = 32 consecutive rows: conflict-free) that is flushed to HBM with
//     contiguous float atomics only when the region moves.  Steps whose box does not fit scatter
//     straight to global memory.
// Recomputes S from Q, K and the bias instead of storing any (M x N) tensor.
// Gradient semantics: see include/bevrender_hip.h (log2-domain inputs as handed in).
#include "attn_tile.h"

namespace {

// table window columns: bf16 80 (40 KiB + 4 x 20 KiB accumulation windows), f32 48 (its staging tiles are 2x larger)
template <int PREC> struct WinCols { static constexpr int value = PREC == BEVR_PREC_BF16 ? 80 : 48; };
constexpr int ACC_PITCH = WIN_PITCH;         // accumulation window rows (floats) per column (rows used: nrows + 1 <= 64)

template <int PREC> struct LdsQ {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int WIN_COLS = WinCols<PREC>::value;
  static constexpr int R_STRIDE = 32 * EB + 16;   // row-layout tiles (K, V): bytes per key row
  static constexpr int T_STRIDE = KT * EB + 16;   // transposed tile (Kt): bytes per channel row
  static constexpr int R_BYTES = KT * R_STRIDE;
  static constexpr int T_BYTES = 32 * T_STRIDE;
  static constexpr int C_BYTES = KT * 16 + 32;
  static constexpr int BUF = 2 * R_BYTES + T_BYTES + C_BYTES;
  static constexpr int WIN = WIN_COLS * WIN_PITCH * 8;
  static constexpr int ACC = (THREADS / 64) * WIN_COLS * ACC_PITCH * 4;   // one window per wave
  static constexpr int TOTAL = 2 * BUF + WIN + ACC;
};

template <int PREC, int NQ>
__global__ __launch_bounds__(THREADS) void attn_bwd_q_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ Kt,
    const char* __restrict__ V, const float* __restrict__ key_a, const float* __restrict__ key_b,
    const char* __restrict__ table_pair, const char* __restrict__ dO, const float* __restrict__ LSE,
    const float* __restrict__ delta, float* __restrict__ dQ, float* __restrict__ dtable) {
  typedef LdsQ<PREC> L;
  constexpr int EB = L::EB;
  constexpr int WIN_COLS = L::WIN_COLS;
  static_assert(L::TOTAL <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* win = smem + 2 * L::BUF;
  float* acc = reinterpret_cast<float*>(smem + 2 * L::BUF + L::WIN) + (threadIdx.x >> 6) * (WIN_COLS * ACC_PITCH);

  const int n_rb = d.Sp / 32;
  const int n_cb = (d.S + 4 * NQ - 1) / (4 * NQ);
  const int n_tile = n_rb * n_cb;
  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / n_tile) * 8 + xcd;
  if (ph >= n_ph) return;
  const int tile = slot % n_tile;
  const int rb = tile % n_rb, cb = tile / n_rb;
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int grp = hd / (d.heads / d.groups);
  const int qb = prob / d.q_div;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const int Mp = d.S * d.Sp;
  const int i0 = rb * 32;

  const char* Qh = Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB;
  const char* dOh = dO + ((size_t)ph * Mp) * 32 * EB;
  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = V + ((size_t)ph * d.Np) * 32 * EB;
  const char* Kth = Kt + ((size_t)ph * 32) * d.Np * EB;
  const float* ka = key_a + (size_t)(prob * d.groups + grp) * d.Np;
  const float* kb = key_b + (size_t)(prob * d.groups + grp) * d.Np;
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  float* dtb = dtable + (size_t)hd * d.Wp * (d.Hp + 1);
  const int Hp8 = d.Hp * 8;
  const int Hq = d.Hp + 1;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int j_first = cb * 4 * NQ;
  const int j_last = min(j_first + 4 * NQ - 1, d.S - 1);
  const float jrx_lo = (float)j_first * rx, jrx_hi = (float)j_last * rx;

  Frag<PREC> qf[NQ], dof[NQ];
  float jrx[NQ], lse[NQ], dlt[NQ];
  int jcol[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    int j = j_first + wave * NQ + t;
    jcol[t] = j;
    const bool live = j < d.S;
    int jc = live ? j : d.S - 1;
    jrx[t] = (float)jc * rx;
    size_t mq = (size_t)jc * d.Sp + i0 + lq;
    qf[t].load(Qh + mq * 32 * EB, hi);
    dof[t].load(dOh + mq * 32 * EB, hi);
    lse[t] = LSE[(size_t)ph * Mp + mq];
    dlt[t] = delta[(size_t)ph * Mp + mq];
    if (!live) {  // duplicate of column S-1: must contribute nothing
      if constexpr (PREC == BEVR_PREC_BF16) {
        dof[t].v[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        dof[t].v[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) dof[t].v[k] = 0.f;
      }
      dlt[t] = 0.f;
    }
  }
  const int ilane = i0 + lq;
  const int rowoff = ilane * 8;
  const int xoffHp = d.x_off * d.Hp;
  const int rot_src = (lane + 63) & 63;

  f32x16 dq[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  constexpr int RCH_ROW = 32 * EB / 16;
  constexpr int TCH_ROW = KT * EB / 16;
  constexpr int NCH = KT * RCH_ROW / THREADS;
  u32x4 stK[NCH], stV[NCH], stT[NCH];
  float st_a = 0.f, st_b = 0.f;
  const int n_step = d.Np / KT;

  auto stage_load = [&](int step) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = tid + c * THREADS;
      stK[c] = *reinterpret_cast<const u32x4*>(Kh + ((size_t)step * KT * RCH_ROW + ch) * 16);
      stV[c] = *reinterpret_cast<const u32x4*>(Vh + ((size_t)step * KT * RCH_ROW + ch) * 16);
      int tr = ch / TCH_ROW, tc = ch % TCH_ROW;
      stT[c] = *reinterpret_cast<const u32x4*>(Kth + ((size_t)tr * d.Np + (size_t)step * KT) * EB + tc * 16);
    }
    if (tid < KT) { st_a = ka[step * KT + tid]; st_b = kb[step * KT + tid]; }
  };
  auto stage_store = [&](int buf, int step) {
    char* base = smem + buf * L::BUF;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = tid + c * THREADS;
      int ro = (ch / RCH_ROW) * L::R_STRIDE + (ch % RCH_ROW) * 16;
      *reinterpret_cast<u32x4*>(base + ro) = stK[c];
      *reinterpret_cast<u32x4*>(base + L::R_BYTES + ro) = stV[c];
      *reinterpret_cast<u32x4*>(base + 2 * L::R_BYTES + (ch / TCH_ROW) * L::T_STRIDE + (ch % TCH_ROW) * 16) = stT[c];
    }
    if (tid < KT) {
      WinInfo wi;
      KeyW kw = stage_keys(st_a, st_b, step * KT + tid < d.N, d, jrx_lo, jrx_hi, WIN_COLS, wi);
      *reinterpret_cast<KeyW*>(base + 2 * L::R_BYTES + L::T_BYTES + tid * 16) = kw;
      if (tid == 0) *reinterpret_cast<WinInfo*>(base + 2 * L::R_BYTES + L::T_BYTES + KT * 16) = wi;
    }
  };

  stage_load(0);
  stage_store(0, 0);
  __syncthreads();

  Region rg;
  rg.ax0 = -(1 << 28);
  rg.ay0 = 0;
  bool acc_live = false;   // the accumulation window holds un-flushed gradient

  // flush the accumulation window of region `r`: per column one contiguous run of 64 floats, non-zeros only
  auto flush = [&](const Region& r) {
    const size_t y0 = (size_t)(i0 + r.ay0 + d.y_off) + lane;
    for (int c = 0; c < WIN_COLS; ++c) {   // this wave's own window
      float v = acc[c * ACC_PITCH + lane];
      if (v != 0.f) atomicAdd(dtb + (size_t)(r.ax0 + c + d.x_off) * Hq + y0, v);
    }
  };

  for (int step = 0; step < n_step; ++step) {
    const int buf = step & 1;
    const char* base = smem + buf * L::BUF;
    if (step + 1 < n_step) stage_load(step + 1);
    const WinInfo wi = *reinterpret_cast<const WinInfo*>(base + 2 * L::R_BYTES + L::T_BYTES + KT * 16);
    const bool use_win = wi.ok != 0;   // workgroup-uniform
    if (use_win && !region_contains(rg, wi, WIN_COLS)) {
      // every wave finished the previous step (barrier at the end of the loop body): safe to drain and move
      if (acc_live) flush(rg);
      rg = region_anchor(wi, d, i0, WIN_COLS);
      load_region(win, tbl, d, rg, i0, WIN_COLS, wave, lane);
      for (int c = 0; c < WIN_COLS; ++c) acc[c * ACC_PITCH + lane] = 0.f;
      acc_live = true;
      __syncthreads();
    }
    const float ax0_f = (float)rg.ax0;
    const int drow = (wi.amin - rg.ay0) + lq;

#pragma unroll
    for (int ks = 0; ks < KT / 32; ++ks) {
      Frag<PREC> kf, vkf, ktf;
      kf.load(base + (ks * 32 + lq) * L::R_STRIDE, hi);
      vkf.load(base + L::R_BYTES + (ks * 32 + lq) * L::R_STRIDE, hi);
      load_perm(ktf, base + 2 * L::R_BYTES + lq * L::T_STRIDE + ks * 32 * EB, hi);
      const KeyW* kc = reinterpret_cast<const KeyW*>(base + 2 * L::R_BYTES + L::T_BYTES) + ks * 32;
      const bool last = (step == n_step - 1) && d.N < d.Np;

#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = -lse[t]; dp[r] = -dlt[t]; }
        s = mma_frag(kf, qf[t], s);       // S^T - LSE
        dp = mma_frag(vkf, dof[t], dp);   // dP^T - delta
        if (use_win) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const KeyW c = kc[crow(r, hi)];
            const float wy0 = 1.0f - c.fy;
            float tx = jrx[t] + (c.b - ax0_f);
            float xf = floorf(tx);
            float fx = tx - xf;
            const int xi = (int)xf;
            const int row = (c.arow8 >> 3) + drow;
            const char* p = win + xi * (WIN_PITCH * 8) + row * 8;
            f32x2 t0 = *reinterpret_cast<const f32x2*>(p);
            f32x2 t1 = *reinterpret_cast<const f32x2*>(p + WIN_PITCH * 8);
            float u0 = t0[0] * wy0 + t0[1] * c.fy;
            float u1 = t1[0] * wy0 + t1[1] * c.fy;
            float sv = s[r] + u0 + fx * (u1 - u0);
            if (last && step * KT + ks * 32 + crow(r, hi) >= d.N) sv = BEVR_NEG_BIG;
            float pr = fast_exp2(sv);
            float ds = BEVR_LN2 * pr * dp[r];
            s[r] = ds;
            float w0 = ds * (1.0f - fx), w1 = ds * fx;
            // Accumulate into this wave's private window with plain read-modify-write (LDS float atomics ran
            // ~3.5x slower here).  A lane's "row + 1" taps are handed to the lane above (rotate by one), so in
            // one pass every active lane owns a distinct row of the two touched columns: 32 rows by the
            // half's own lanes, the 33rd by the first lane of the other half.  The two halves hold different
            // keys that may share cells (the projector pins every out-of-image key to one pixel), so they
            // go in two passes; a wave's LDS operations execute in order, which sequences the passes.
            const float c00 = w0 * wy0, c01 = w0 * c.fy, c10 = w1 * wy0, c11 = w1 * c.fy;
            const int goff = xi * ACC_PITCH + row;
            const float u01 = __shfl(c01, rot_src), u11 = __shfl(c11, rot_src);
            const int ugoff = __shfl(goff, rot_src) + 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bool mine = hi == h;
              const bool top = lane == (h == 0 ? 32 : 0);
              if (mine || top) {
                float* g = acc + (mine ? goff : ugoff);
                const float a0 = mine ? (lq ? c00 + u01 : c00) : u01;
                const float a1 = mine ? (lq ? c10 + u11 : c10) : u11;
                g[0] += a0;
                g[ACC_PITCH] += a1;
              }
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const KeyW c = kc[crow(r, hi)];
            const float wy0 = 1.0f - c.fy;
            float tx = jrx[t] + c.b;
            float xf = floorf(tx);
            float fx = tx - xf;
            int xi = (int)xf;
            unsigned off = (unsigned)(xi * Hp8 + c.aoff + rowoff);
            f32x2 t0 = *reinterpret_cast<const f32x2*>(tbl + off);
            f32x2 t1 = *reinterpret_cast<const f32x2*>(tbl + off + Hp8);
            float u0 = t0[0] * wy0 + t0[1] * c.fy;
            float u1 = t1[0] * wy0 + t1[1] * c.fy;
            float sv = s[r] + u0 + fx * (u1 - u0);
            if (last && step * KT + ks * 32 + crow(r, hi) >= d.N) sv = BEVR_NEG_BIG;
            float pr = fast_exp2(sv);
            float ds = BEVR_LN2 * pr * dp[r];
            s[r] = ds;
            if (ds != 0.f) {
              // plain transposed table, row pitch Hp + 1
              int yi = (c.aoff >> 3) - xoffHp + ilane;
              float* g0 = dtb + (size_t)(xi + d.x_off) * Hq + yi;
              float w0 = ds * (1.0f - fx), w1 = ds * fx;
              atomicAdd(g0, w0 * wy0);
              atomicAdd(g0 + 1, w0 * c.fy);
              atomicAdd(g0 + Hq, w1 * wy0);
              atomicAdd(g0 + Hq + 1, w1 * c.fy);
            }
          }
        }
        dq[t] = mma_acc_b(ktf, s, dq[t]);
      }
    }

    if (step + 1 < n_step) stage_store(buf ^ 1, step + 1);
    __syncthreads();
  }
  if (acc_live) flush(rg);

  float* dQh = dQ + ((size_t)ph * Mp) * 32;
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    if (jcol[t] >= d.S) continue;
    size_t mq = (size_t)jcol[t] * d.Sp + i0 + lq;
    float* row = dQh + mq * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = dq[t][4 * g4 + k];
      *reinterpret_cast<f32x4*>(row + 8 * g4 + 4 * hi) = v;
    }
  }
}

template <int PREC, int NQ>
int launch(const bevr_attn_desc& d, const void* Q, const void* K, const void* Kt, const void* V, const float* key_a,
           const float* key_b, const float* table_pair, const void* dO, const float* LSE, const float* delta,
           float* dQ, float* dtable, hipStream_t st) {
  const int n_rb = d.Sp / 32, n_cb = (d.S + 4 * NQ - 1) / (4 * NQ);
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_rb * n_cb;
  const size_t lds = LdsQ<PREC>::TOTAL;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<PREC, NQ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((attn_bwd_q_kernel<PREC, NQ>), dim3(grid), dim3(THREADS), lds, st, d, (const char*)Q,
                     (const char*)K, (const char*)Kt, (const char*)V, key_a, key_b, (const char*)table_pair,
                     (const char*)dO, LSE, delta, dQ, dtable);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_bwd_q(const bevr_attn_desc* d, const void* Q, const void* K, const void* Kt, const void* V,
                               const float* key_a, const float* key_b, const float* table_pair, const void* dO,
                               const float* LSE, const float* delta, float* dQ, float* dtable, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !K || !Kt || !V || !key_a || !key_b || !table_pair || !dO || !LSE || !delta || !dQ || !dtable)
    return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(K) || !bevr_aligned16(Kt) || !bevr_aligned16(V) || !bevr_aligned16(dO) ||
      !bevr_aligned16(dQ) || !bevr_aligned16(table_pair))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16, 2>(*d, Q, K, Kt, V, key_a, key_b, table_pair, dO, LSE, delta, dQ, dtable, st);
  return launch<BEVR_PREC_F32, 2>(*d, Q, K, Kt, V, key_a, key_b, table_pair, dO, LSE, delta, dQ, dtable, st);
}
