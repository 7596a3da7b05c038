// Fused BEV attention forward for gfx950:  O = softmax_n( Q^T K * scale + bilinear_rpe_bias ) V
// without materialising the (M x N) logits, bias or displacement tensors the reference builds
// (model/SCA_deform_attn.py:331-413, model/TSA_deform_attn.py:245-333).
//
// Orientation ("query on the lane"): each 32x32 MFMA tile is S^T[key][query] -- the accumulator row
// (register) is the key, the column (lane) is the query.  Consequences:
//   * softmax statistics of a query are lane-local (16 registers + one cross-half exchange);
//   * P^T is already the B operand of O^T[ch][q] += V^T[ch][key] P^T[key][q]  (no LDS transpose);
//   * the lanes of a tile are 32 consecutive BEV rows i of one BEV column j, so with the rpe table
//     stored transposed (y contiguous) the bilinear taps of the 32 lanes are consecutive addresses:
//     conflict-free 8-byte LDS reads from the step's table window (attn_tile.h), or two coalesced
//     256-byte global rows on the fallback path.
// Work split: workgroup = 4 waves = one 32-row block x (4*NQ) BEV columns; wave w owns NQ columns;
// all waves walk the keys together, 64 per step, K / V^T / key constants staged through LDS
// (double buffered).  blockIdx is remapped so that all query tiles of one (problem, head) run on one
// XCD and stream the same K/V through that XCD's L2.
#include "attn_tile.h"

namespace {

constexpr float RESCALE_THR = 8.0f;  // log2 units: lazy running-max update (P <= 2^8)
constexpr int WIN_COLS = 96;         // window columns held in LDS (x 64 rows x 8 B = 48 KiB)

template <int PREC> struct Lds {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int K_STRIDE = 32 * EB + 16;   // bytes per key row (+16: bank spread)
  static constexpr int V_STRIDE = KT * EB + 16;   // bytes per channel row
  static constexpr int K_BYTES = KT * K_STRIDE;
  static constexpr int V_BYTES = 32 * V_STRIDE;
  static constexpr int C_BYTES = KT * 16 + 32;    // KeyW per key + WinInfo
  static constexpr int BUF = K_BYTES + V_BYTES + C_BYTES;
  static constexpr int WIN = WIN_COLS * WIN_PITCH * 8;
  static constexpr int TOTAL = 2 * BUF + WIN;
};

template <int PREC, int NQ>
__global__ __launch_bounds__(THREADS, PREC == BEVR_PREC_BF16 ? 2 : 1) void attn_fwd_kernel(bevr_attn_desc d, const char* __restrict__ Q,
                                                              const char* __restrict__ K,
                                                              const char* __restrict__ Vt,
                                                              const float* __restrict__ key_a,
                                                              const float* __restrict__ key_b,
                                                              const char* __restrict__ table_pair,
                                                              float* __restrict__ O, float* __restrict__ LSE) {
  typedef Lds<PREC> L;
  constexpr int EB = L::EB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* win = smem + 2 * L::BUF;

  // ---- which (problem, head, query tile) -------------------------------------------------------
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
  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = Vt + ((size_t)ph * 32) * d.Np * EB;
  const float* ka = key_a + (size_t)(prob * d.groups + grp) * d.Np;
  const float* kb = key_b + (size_t)(prob * d.groups + grp) * d.Np;
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const int Hp8 = d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int j_first = cb * 4 * NQ;
  const int j_last = min(j_first + 4 * NQ - 1, d.S - 1);
  const float jrx_lo = (float)j_first * rx, jrx_hi = (float)j_last * rx;

  // ---- per-wave query columns -------------------------------------------------------------------
  Frag<PREC> qf[NQ];
  float jrx[NQ];
  int jcol[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    int j = j_first + wave * NQ + t;
    jcol[t] = j;
    int jc = j < d.S ? j : d.S - 1;  // columns past the grid: compute on a clamped copy, never stored
    jrx[t] = (float)jc * rx;
    qf[t].load(Qh + ((size_t)jc * d.Sp + i0 + lq) * 32 * EB, hi);
  }
  const int rowoff = (i0 + lq) * 8;
  const int lq8 = lq * 8;

  f32x16 o[NQ];
  float m[NQ], l[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    m[t] = BEVR_NEG_BIG;
    l[t] = 0.f;
  }

  // ---- staging: global -> registers -> LDS ------------------------------------------------------
  constexpr int KCH_ROW = 32 * EB / 16;            // 16-B chunks per K row
  constexpr int VCH_ROW = KT * EB / 16;            // 16-B chunks per V^T row (this step's keys)
  constexpr int NCH = KT * KCH_ROW / THREADS;      // chunks per thread for each of K and V (1 or 2)
  static_assert(KT * KCH_ROW % THREADS == 0 && 32 * VCH_ROW == KT * KCH_ROW, "staging shape");
  u32x4 stK[NCH], stV[NCH];
  float st_a = 0.f, st_b = 0.f;
  const int n_step = d.Np / KT;

  auto stage_load = [&](int step) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = tid + c * THREADS;
      stK[c] = *reinterpret_cast<const u32x4*>(Kh + ((size_t)step * KT * KCH_ROW + ch) * 16);
      int vr = ch / VCH_ROW, vc = ch % VCH_ROW;
      stV[c] = *reinterpret_cast<const u32x4*>(Vh + ((size_t)vr * d.Np + (size_t)step * KT) * EB + vc * 16);
    }
    if (tid < KT) { st_a = ka[step * KT + tid]; st_b = kb[step * KT + tid]; }
  };
  auto stage_store = [&](int buf, int step) {
    char* base = smem + buf * L::BUF;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = tid + c * THREADS;
      *reinterpret_cast<u32x4*>(base + (ch / KCH_ROW) * L::K_STRIDE + (ch % KCH_ROW) * 16) = stK[c];
      *reinterpret_cast<u32x4*>(base + L::K_BYTES + (ch / VCH_ROW) * L::V_STRIDE + (ch % VCH_ROW) * 16) = stV[c];
    }
    if (tid < KT) {   // exactly wave 0
      WinInfo wi;
      KeyW kw = stage_keys(st_a, st_b, step * KT + tid < d.N, d, jrx_lo, jrx_hi, WIN_COLS, wi);
      *reinterpret_cast<KeyW*>(base + L::K_BYTES + L::V_BYTES + tid * 16) = kw;
      if (tid == 0) *reinterpret_cast<WinInfo*>(base + L::K_BYTES + L::V_BYTES + KT * 16) = wi;
    }
  };

  stage_load(0);
  stage_store(0, 0);
  __syncthreads();

  Region rg;
  rg.ax0 = -(1 << 28);   // nothing contained: first windowed step anchors
  rg.ay0 = 0;

  for (int step = 0; step < n_step; ++step) {
    const int buf = step & 1;
    const char* base = smem + buf * L::BUF;
    if (step + 1 < n_step) stage_load(step + 1);
    const WinInfo wi = *reinterpret_cast<const WinInfo*>(base + L::K_BYTES + L::V_BYTES + KT * 16);
    const bool use_win = wi.ok != 0;   // workgroup-uniform
    if (use_win && !region_contains(rg, wi, WIN_COLS)) {
      rg = region_anchor(wi, d, i0, WIN_COLS);
      load_region(win, tbl, d, rg, i0, WIN_COLS, THREADS / 64, wave, lane);
      __syncthreads();
    }
    const float ax0_f = (float)rg.ax0;
    const int drow8 = (wi.amin - rg.ay0) * 8 + lq8;

#pragma unroll
    for (int ks = 0; ks < KT / 32; ++ks) {
      Frag<PREC> kf, vf;
      kf.load(base + (ks * 32 + lq) * L::K_STRIDE, hi);
      load_perm(vf, base + L::K_BYTES + lq * L::V_STRIDE + ks * 32 * EB, hi);
      const KeyW* kc = reinterpret_cast<const KeyW*>(base + L::K_BYTES + L::V_BYTES) + ks * 32;

      f32x16 s[NQ];
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        s[t] = mma_frag(kf, qf[t], s[t]);
      }

      // relative-position bias: rows of the tile are keys crow(r, hi); lanes are BEV rows i0 + lq.
      if (use_win) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const KeyW c = kc[crow(r, hi)];
          const float wy0 = 1.0f - c.fy;
          const float bl = c.b - ax0_f;
          const char* wr = win + c.arow8 + drow8;
#pragma unroll
          for (int t = 0; t < NQ; ++t) {
            float tx = jrx[t] + bl;
            float xf = floorf(tx);
            float fx = tx - xf;
            const char* p = wr + (int)xf * (WIN_PITCH * 8);
            f32x2 t0 = *reinterpret_cast<const f32x2*>(p);
            f32x2 t1 = *reinterpret_cast<const f32x2*>(p + WIN_PITCH * 8);
            float u0 = t0[0] * wy0 + t0[1] * c.fy;
            float u1 = t1[0] * wy0 + t1[1] * c.fy;
            s[t][r] += u0 + fx * (u1 - u0);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const KeyW c = kc[crow(r, hi)];
          const float wy0 = 1.0f - c.fy;
          const int ar = c.aoff + rowoff;
#pragma unroll
          for (int t = 0; t < NQ; ++t) {
            float tx = jrx[t] + c.b;
            float xf = floorf(tx);
            float fx = tx - xf;
            unsigned off = (unsigned)((int)xf * Hp8 + ar);
            f32x2 t0 = *reinterpret_cast<const f32x2*>(tbl + off);
            f32x2 t1 = *reinterpret_cast<const f32x2*>(tbl + off + Hp8);
            float u0 = t0[0] * wy0 + t0[1] * c.fy;
            float u1 = t1[0] * wy0 + t1[1] * c.fy;
            s[t][r] += u0 + fx * (u1 - u0);
          }
        }
      }
      // mask padded keys (only the last step can hold any)
      if (step == n_step - 1 && d.N < d.Np) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          bool dead = step * KT + ks * 32 + crow(r, hi) >= d.N;
#pragma unroll
          for (int t = 0; t < NQ; ++t) s[t][r] = dead ? BEVR_NEG_BIG : s[t][r];
        }
      }

#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        float tm = s[t][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, s[t][r]);
        tm = fmaxf(tm, __shfl_xor(tm, 32));
        if (__any(tm > m[t] + RESCALE_THR)) {   // wave-uniform: rare after the first tiles
          float mn = fmaxf(m[t], tm);
          float al = fast_exp2(m[t] - mn);
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= al;
          l[t] *= al;
          m[t] = mn;
        }
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = fast_exp2(s[t][r] - m[t]);
          s[t][r] = p;
          ls += p;
        }
        l[t] += ls;
        o[t] = mma_acc_b(vf, s[t], o[t]);
      }
    }

    if (step + 1 < n_step) stage_store(buf ^ 1, step + 1);
    __syncthreads();
  }

  // ---- epilogue: normalise, store O^T tile as [q][32] rows and the log2-sum-exp ------------------
  float* Oh = O + ((size_t)ph * Mp) * 32;
  float* Lh = LSE + (size_t)ph * Mp;
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    if (jcol[t] >= d.S) continue;
    float lt = l[t] + __shfl_xor(l[t], 32);
    float inv = 1.0f / lt;
    size_t mq = (size_t)jcol[t] * d.Sp + i0 + lq;
    float* orow = Oh + mq * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = o[t][4 * g4 + k] * inv;
      *reinterpret_cast<f32x4*>(orow + 8 * g4 + 4 * hi) = v;
    }
    if (hi == 0) Lh[mq] = m[t] + __log2f(lt);
  }
}

template <int PREC, int NQ>
int launch_fwd(const bevr_attn_desc& d, const void* Q, const void* K, const void* Vt, const float* key_a,
               const float* key_b, const float* table_pair, float* O, float* LSE, hipStream_t st) {
  const int n_rb = d.Sp / 32, n_cb = (d.S + 4 * NQ - 1) / (4 * NQ);
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_rb * n_cb;
  const size_t lds = Lds<PREC>::TOTAL;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<PREC, NQ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<PREC, NQ>), dim3(grid), dim3(THREADS), lds, st, d, (const char*)Q,
                     (const char*)K, (const char*)Vt, key_a, key_b, (const char*)table_pair, O, LSE);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_fwd(const bevr_attn_desc* d, const void* Q, const void* K, const void* Vt,
                             const float* key_a, const float* key_b, const float* table_pair, float* O,
                             float* LSE, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !K || !Vt || !key_a || !key_b || !table_pair || !O || !LSE) return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(K) || !bevr_aligned16(Vt) || !bevr_aligned16(O) ||
      !bevr_aligned16(table_pair))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch_fwd<BEVR_PREC_BF16, 2>(*d, Q, K, Vt, key_a, key_b, table_pair, O, LSE, st);
  return launch_fwd<BEVR_PREC_F32, 2>(*d, Q, K, Vt, key_a, key_b, table_pair, O, LSE, st);
}
