// Key-side backward of the tap kernels (attn_tap.h): the gradient of every key's POSITION -- its rpe-table coordinates
// (a, b) through the bias and its sampling position (ys, xs) through the tap weights.  With P and dS as in
// attn_tap_bwd_q.hip (dS = P (w . H + Hc), ln2 and delta folded into H by the caller),
//     dw_t(n) = sum_q dS[n][q] G[t][q] + P[n][q] H[t][q] / ln2  (the logit path and the value path V_n = sum_t w_t Vpix_t)
//     Z[cell][n] = sum_q dS[n][q] Tsh[cell][q]                  per BEV column: the chunk's cells are a different part of the
//                                                                table for every column
//     d a_n = sum_c wx_c (Z[c][r0 + 1] - Z[c][r0]),   d b_n = sum_r wy_r (Z[c0 + 1][r] - Z[c0][r])     (bilinear derivative)
// (reference: autograd through model/SCA_deform_attn.py:290-301 and :365-394).  No dK / dV exists: the feature-map and
// projection-weight gradients follow from dG (bwd_q) and the forward's R by thin GEMMs in the caller.
//
// Key-stationary: a wave owns one 32-key tile (two 16-key matrix tiles), a workgroup 7 tiles + a PRODUCER wave that
// stages the query side -- the rows G[q][.], H[q][.] of two 32-row slabs of one BEV column per step -- through LDS; one
// barrier per step.  S[q][n] comes out with the key on the lane, so P and dS are the B operands of the Z products as they
// stand; the transposed query-side operands (G^T, H^T) come out of the staged rows through ds_read_b64_tr_b16.
// The table: per column the producer copies the 8 columns x (S + 16) rows the workgroup's chunks can touch into a SHARED
// LDS window (three of them in rotation, filled one column ahead in slices), as 16-bit hi and lo parts (Z is differenced
// in d a / d b: with the hi part alone the position gradient carries the table's 2^-9 rounding) and
// in two row-parity copies (4 consecutive rows are one aligned
// ds_read2_b32 for any first row).  A tile whose taps do not fit one chunk for a column takes the per-pair gather for
// that column (any key set is handled; a cell-sorted segment never does).
#include <type_traits>
#include "attn_tap.h"

namespace {

constexpr int NKW = 7;                 // key waves per workgroup (A/B on the benchmark shape: 3, 5 and 9 are 27-45 % slower)
constexpr int NCW = 8;                 // table columns of the shared window (a chunk is 4 wide: origins may differ by 4 columns ...
constexpr int NRX = 8;                 // ... and by 8 rows inside one workgroup)
constexpr int SPB = 2;                 // 32-row slabs per step (= per barrier; one per barrier: 27.5 against 24.4 ms)
struct LdsK {
  static constexpr int OFF_G = 0;      // [32 rows][16 slots] 16-bit
  static constexpr int OFF_H = 1024;
  static constexpr int SLAB = 2048;    // one slab's G | H rows
  static constexpr int BUF = SPB * SLAB;
};
// dwords per (kind, parity, column) of a window: rows 0 .. Sp + NRX + 7, two rows per dword
__host__ __device__ __forceinline__ int win_dwords(int Sp) { return (Sp + NRX + 8) / 2; }

// does tile box `sb` take the matrix path in BEV column j?  Its taps fit one chunk, and the chunk lies inside the
// workgroup's window (origin row a0w, leftmost coordinate bminw)
__device__ __forceinline__ bool tile_fits(const StepBox& sb, float jrx, int a0w, float bminw) {
  const int da = sb.amin - a0w;
  const int dx = (int)floorf(jrx + sb.bmin) - (int)floorf(jrx + bminw);
  return sb.amax >= sb.amin && da >= 0 && da <= NRX && box_fits(sb, jrx) && dx >= 0 && dx <= NCW - CELL_C;
}

// SLOW = false: the (tile, column) pairs that take the matrix path; SLOW = true: the others, by the per-pair gather (a
// workgroup without any leaves at once: every workgroup of a cell-sorted segment).  Two launches: with both bodies in one
// loop the matrix path reloaded loop invariants from scratch on every slab.
template <int PREC, bool SLOW>
__global__ __launch_bounds__(64 * (NKW + 1), 4) void attn_tap_bwd_k_kernel(
    bevr_attn_desc d, const char* __restrict__ G, const char* __restrict__ H, const char* __restrict__ tap_ws,
    const float* __restrict__ table_t, float* __restrict__ dkey_a, float* __restrict__ dkey_b,
    float* __restrict__ dkey_y, float* __restrict__ dkey_x, int n_wg_ph) {
  typedef LdsK L;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / n_wg_ph) * 8 + xcd;
  if (ph >= n_ph) return;
  const int wg = slot % n_wg_ph;
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int Mp = d.S * d.Sp;
  const int nslab = d.Sp / 32, nstep = (nslab + SPB - 1) / SPB;
  const int n_tiles = d.Np / 32;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int HpT = d.Hp + 1;                                   // rows of a column of the plain transposed table
  const float* tbl = table_t + (size_t)hd * d.Wp * HpT;
  const StepBox* box = reinterpret_cast<const StepBox*>(tap_ws + tap_ws_box_offset(d)) + (size_t)prob * (d.Np / 32);
  const int NRWD = win_dwords(d.Sp);
  const int win_bytes = 4 * NCW * NRWD * 4;                   // [kind hi / lo][row parity][NCW columns][NRWD dwords]
  char* win_base = smem + 2 * L::BUF;
  // the key waves' records live in LDS between their uses (column start, column end): 8 registers less in the slab loop
  TapRec* stash = reinterpret_cast<TapRec*>(win_base + 3 * win_bytes) + wave * 32;

  // the workgroup's window origin: the lowest table row / leftmost coordinate of its tiles (rows: for every column)
  int a0w = 0x7fffffff;
  float bminw = 3.0e38f;
  for (int t = 0; t < NKW; ++t) {
    const int tile = wg * NKW + t;
    if (tile < n_tiles) {
      const StepBox b = box[tile];
      if (b.amax >= b.amin) {
        a0w = min(a0w, b.amin);
        bminw = fminf(bminw, b.bmin);
      }
    }
  }
  if (a0w == 0x7fffffff) { a0w = 0; bminw = 0.f; }
  if constexpr (SLOW) {
    bool unfit = false;     // the same for every wave of the workgroup: all of them leave, or none
    for (int t = 0; t < NKW; ++t) {
      const int tile = wg * NKW + t;
      if (tile >= n_tiles) break;
      const StepBox b = box[tile];
      if (b.amax < b.amin) continue;
      for (int j = lane; j < d.S; j += 64) unfit = unfit || !tile_fits(b, (float)j * rx, a0w, bminw);
    }
    if (!__any(unfit)) return;
  }

  if (wave == NKW) {
    // ---- producer: the G and H rows of slab (j, i0) -> LDS one step ahead of the key waves, and the table window of the
    // NEXT column in slices, one per slab of the current one.  THREE windows: while the slices of column j + 1 are written
    // the key waves may still be one step behind, in the last slab of column j - 1 ---------------------------------------
    __builtin_amdgcn_s_setprio(3);
    const int n_quad = (NRWD + 1) / 2, n_item = NCW * n_quad;
    // item = (column c, rows 4 m .. 4 m + 4) of the window of BEV column jn: five table values in, hi / lo parts of the row
    // pairs (4 m, + 1), (+ 2, + 3) [parity 0] and (+ 1, + 2), (+ 3, + 4) [parity 1] out.  Load and store are separate
    // steps: the loads of one slice are in flight across a barrier (they sat on the workgroup's critical path when the
    // producer waited for them before every barrier)
    auto item_load = [&](int jn, int it, float (&t)[5]) {
      if (it >= n_item || jn >= d.S) return;
      const int c = it / n_quad, m = it - c * n_quad;
      const int x0w = (int)floorf((float)jn * rx + bminw);
      const int xc = max(0, min(x0w + c + d.x_off, d.Wp - 1));
      // rows past the padded table are never weighted: clamp the run's start (the table's last rows are zero padding)
      const int y = max(0, min(a0w + d.y_off + 4 * m, HpT - 5));
      const float* src = tbl + (size_t)xc * HpT + y;
#pragma unroll
      for (int k = 0; k < 5; ++k) t[k] = src[k];
    };
    auto item_store = [&](int jn, int it, const float (&t)[5]) {
      if (it >= n_item || jn >= d.S) return;
      const int c = it / n_quad, m = it - c * n_quad;
      const uint32_t h01 = Half<PREC>::pack2(t[0], t[1]), h23 = Half<PREC>::pack2(t[2], t[3]);
      const uint32_t h12 = Half<PREC>::pack2(t[1], t[2]), h34 = Half<PREC>::pack2(t[3], t[4]);
      const uint32_t l01 = Half<PREC>::pack2(t[0] - Half<PREC>::lo(h01), t[1] - Half<PREC>::hi(h01));
      const uint32_t l23 = Half<PREC>::pack2(t[2] - Half<PREC>::lo(h23), t[3] - Half<PREC>::hi(h23));
      const uint32_t l12 = Half<PREC>::pack2(t[1] - Half<PREC>::lo(h12), t[2] - Half<PREC>::hi(h12));
      const uint32_t l34 = Half<PREC>::pack2(t[3] - Half<PREC>::lo(h34), t[4] - Half<PREC>::hi(h34));
      uint32_t* w = reinterpret_cast<uint32_t*>(win_base + (jn % 3) * win_bytes) + c * NRWD + 2 * m;
      const bool second = 2 * m + 1 < NRWD;
      w[0] = h01;
      w[NCW * NRWD] = h12;
      w[2 * NCW * NRWD] = l01;
      w[3 * NCW * NRWD] = l12;
      if (second) {
        w[1] = h23;
        w[NCW * NRWD + 1] = h34;
        w[2 * NCW * NRWD + 1] = l23;
        w[3 * NCW * NRWD + 1] = l34;
      }
    };
    // the slice of the next column's window that step (j, s) writes: items [s per_step, (s + 1) per_step), two per lane
    const int per_step = (n_item + nstep - 1) / nstep;
    auto slice_item = [&](int s2, int r) {
      const int k = r * 64 + lane;
      return k < per_step ? s2 * per_step + k : n_item;
    };
    float f0[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, f1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (!SLOW) {
      for (int it = lane; it < n_item; it += 64) {   // column 0's window, whole, before the first step
        float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        item_load(0, it, t);
        item_store(0, it, t);
      }
      item_load(1, slice_item(0, 0), f0);
      item_load(1, slice_item(0, 1), f1);
    }
    const char* Gp = G + ((size_t)ph * Mp) * 32 + lane * 16;
    const char* Hq = H + ((size_t)ph * Mp) * 32 + lane * 16;
    // slab `sl` (0 .. S nslab - 1, column-major) of the packed rows starts sl * 1024 bytes in; a step stages SPB slabs
    auto slab_index = [&](int j2, int st2, int u) { return min(j2 * nslab + min(SPB * st2 + u, nslab - 1), d.S * nslab - 1); };
    int e = 0;
    u32x4 gv[SPB], hv[SPB];
#pragma unroll
    for (int u = 0; u < SPB; ++u) {
      gv[u] = gload16(Gp + (size_t)slab_index(0, 0, u) * 1024);
      hv[u] = gload16(Hq + (size_t)slab_index(0, 0, u) * 1024);
    }
    for (int j = 0; j < d.S; ++j) {
      for (int st = 0; st < nstep; ++st, ++e) {
        char* bb = smem + (e & 1) * L::BUF;
#pragma unroll
        for (int u = 0; u < SPB; ++u) {
          *reinterpret_cast<u32x4*>(bb + u * L::SLAB + L::OFF_G + lane * 16) = gv[u];
          *reinterpret_cast<u32x4*>(bb + u * L::SLAB + L::OFF_H + lane * 16) = hv[u];
        }
        if constexpr (!SLOW) {
          item_store(j + 1, slice_item(st, 0), f0);
          item_store(j + 1, slice_item(st, 1), f1);
          // a slice beyond the two pipelined items per lane (short columns: few steps share the window): on the spot
          for (int k = 128 + lane; k < per_step; k += 64) {
            float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            item_load(j + 1, st * per_step + k, t);
            item_store(j + 1, st * per_step + k, t);
          }
        }
        // the next step's rows and the next window slice: in flight across the barrier
        const int jn = st + 1 < nstep ? j : min(j + 1, d.S - 1), stn = st + 1 < nstep ? st + 1 : 0;
#pragma unroll
        for (int u = 0; u < SPB; ++u) {
          gv[u] = gload16(Gp + (size_t)slab_index(jn, stn, u) * 1024);
          hv[u] = gload16(Hq + (size_t)slab_index(jn, stn, u) * 1024);
        }
        if constexpr (!SLOW) {
          const int jf = st + 1 < nstep ? j + 1 : j + 2;
          item_load(jf, slice_item(stn, 0), f0);
          item_load(jf, slice_item(stn, 1), f1);
        }
        __syncthreads();
      }
    }
    __syncthreads();
    return;
  }

  // ---- key waves ---------------------------------------------------------------------------------------------------
  const int tile = wg * NKW + wave;
  const bool have_tile = tile < n_tiles;            // uniform; a wave without a tile only keeps the barriers
  const TapRec* recs = reinterpret_cast<const TapRec*>(tap_ws) + (size_t)prob * d.Np;
  const int tl = have_tile ? tile : 0;
  const StepBox sb = box[tl];
  const int da = sb.amin - a0w;                       // rows between the window's origin and this tile's chunk origin

  bf16x8 bk[2];        // B operand of S / dP for key sub-tile kb: lanes 0..31 the tap slots, lanes 32..63 the cells (per column)
  // Z[slot 4 g + e][key] of the logit path (G^T dS) and of the value path (H^T P: H carries ln2, undone at the end),
  // Z[cell (c = g, r = e)][key]
  f32x4 zt[2], zv[2], zc[2];
  float acc_a[2] = {0.f, 0.f}, acc_b[2] = {0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const TapRec r0 = recs[(size_t)tl * 32 + 16 * kb + li];
    if (g == 0) stash[16 * kb + li] = r0;
    u32x4 t0, t1;
    tap_weights<PREC>(r0.ys, r0.xs, t0, t1);
    bk[kb] = __builtin_bit_cast(bf16x8, g == 0 ? t0 : t1);
    zt[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    zv[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    zc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int a_row = L::OFF_G + li * 32 + (g & 1) * 16;                          // row read of a staged image (lanes 0..31)
  const int t_off = (4 * g + (li >> 2)) * 32 + (lane & 3) * 8;                  // transposed read, second block + 512
  const int cell_c = li >> 2, cell_r = li & 3;                                   // this lane's cell in the transposed table operands

  int e = 0;
  float jrx = 0.f;
  // this column's window and this lane's byte offsets into it at slab 0 (a row is 2 bytes, so slab i0 is 2 i0 further; the
  // row parity of a lane's reads does not change from slab to slab): the plain cell reads of the S operand (columns
  // 2 (g - 2) and + 1) and the transposed reads of the Z operands (hi and lo parts)
  const char* win_b = nullptr;
  int wq0 = 0, wq1 = 0, wth = 0, wtl = 0;
  // one 32-row slab of column j against this wave's 32 keys.  FIT: the tile's taps fit one chunk inside the window (bias
  // and its position gradient through the matrix cores); else the per-pair gather from the table in global memory
  auto slab = [&](auto fit_tag, const char* base, int i0) {
    constexpr bool FIT = decltype(fit_tag)::value;
    // (rows past the grid carry the offset -big in G: their weights are 0 without a test here)
    bf16x8 qa[2], ha[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      u32x4 v = {0u, 0u, 0u, 0u}, hv = {0u, 0u, 0u, 0u};
      if (g < 2) {
        v = *reinterpret_cast<const u32x4*>(base + a_row + rb * 512);
        hv = *reinterpret_cast<const u32x4*>(base + L::OFF_H - L::OFF_G + a_row + rb * 512);
      } else if (FIT) {
        // the chunk's cells for BEV row i0 + 16 rb + li: columns 2 (g - 2), + 1; rows i .. i + 3 of the window (addresses
        // prepared per column: wq0 / wq1; a slab is 16 dwords further, the second row block 8 more)
        const uint32_t* w0 = reinterpret_cast<const uint32_t*>(win_b + wq0 + 2 * i0) + 8 * rb;
        const uint32_t* w1 = reinterpret_cast<const uint32_t*>(win_b + wq1 + 2 * i0) + 8 * rb;
        v[0] = w0[0]; v[1] = w0[1];
        v[2] = w1[0]; v[3] = w1[1];
      }
      qa[rb] = __builtin_bit_cast(bf16x8, v);
      ha[rb] = __builtin_bit_cast(bf16x8, hv);
    }
    const bf16x8 gt = lds_tr8(base + L::OFF_G + t_off, 512);
    const bf16x8 ht = lds_tr8(base + L::OFF_H + t_off, 512);
    bf16x8 thi = gt, tlo = gt;
    if constexpr (FIT) {
      const uint32_t* w0 = reinterpret_cast<const uint32_t*>(win_b + wth + 2 * i0);
      const uint32_t* w1 = reinterpret_cast<const uint32_t*>(win_b + wtl + 2 * i0);
      u32x4 a, b;
      a[0] = w0[0]; a[1] = w0[1]; a[2] = w0[8]; a[3] = w0[9];
      b[0] = w1[0]; b[1] = w1[1]; b[2] = w1[8]; b[3] = w1[9];
      thi = __builtin_bit_cast(bf16x8, a);
      tlo = __builtin_bit_cast(bf16x8, b);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 s0 = mfma16<PREC>(qa[0], bk[kb], z4);       // S[query 4 g + e of row block 0][key]
      f32x4 s1 = mfma16<PREC>(qa[1], bk[kb], z4);
      const f32x4 q0 = mfma16<PREC>(ha[0], bk[kb], z4);  // dP
      const f32x4 q1 = mfma16<PREC>(ha[1], bk[kb], z4);
      float p[8], ds[8];
      if constexpr (!FIT) {
        const TapRec rk = stash[16 * kb + li];
        const float a = rk.a, tx = jrx + rk.b;
        const float af = floorf(a), xf = floorf(tx);
        const float fy = a - af, fx = tx - xf;
        const bool dead = rk.ys < -50.0f;
        const int xc = max(0, min((int)xf + d.x_off, d.Wp - 2));
        const int yb = (int)af + d.y_off + i0 + 4 * g;
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int y = max(0, min(yb + (r & 3) + 16 * (r >> 2), HpT - 2));
          const float* c0p = tbl + (size_t)xc * HpT + y;
          const float t00 = c0p[0], t01 = c0p[1], t10 = c0p[HpT], t11 = c0p[HpT + 1];
          const float u0 = t00 + fy * (t01 - t00), u1 = t10 + fy * (t11 - t10);
          const float sv = (r < 4 ? s0[r & 3] : s1[r & 3]) + (dead ? 0.f : u0 + fx * (u1 - u0));
          p[r] = fast_exp2(sv);
          ds[r] = p[r] * (r < 4 ? q0[r & 3] : q1[r & 3]);
          pa += ds[r] * ((1.0f - fx) * (t01 - t00) + fx * (t11 - t10));
          pb += ds[r] * (u1 - u0);
        }
        acc_a[kb] += dead ? 0.f : pa;
        acc_b[kb] += dead ? 0.f : pb;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = fast_exp2(s0[r]);
          ds[r] = p[r] * q0[r];
          p[4 + r] = fast_exp2(s1[r]);
          ds[4 + r] = p[4 + r] * q1[r];
        }
      }
      u32x4 dsw, pw;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dsw[k] = Half<PREC>::pack2(ds[2 * k], ds[2 * k + 1]);
        pw[k] = Half<PREC>::pack2(p[2 * k], p[2 * k + 1]);
      }
      const bf16x8 ds8 = __builtin_bit_cast(bf16x8, dsw), p8 = __builtin_bit_cast(bf16x8, pw);
      zt[kb] = mfma16<PREC>(gt, ds8, zt[kb]);
      zv[kb] = mfma16<PREC>(ht, p8, zv[kb]);
      if constexpr (FIT) {
        zc[kb] = mfma16<PREC>(tlo, ds8, zc[kb]);
        zc[kb] = mfma16<PREC>(thi, ds8, zc[kb]);
      }
    }
  };

  for (int j = 0; j < d.S; ++j) {
    jrx = (float)j * rx;
    const int x0 = (int)floorf(jrx + sb.bmin), a0 = sb.amin;
    const int dx = x0 - (int)floorf(jrx + bminw);
    const bool fit = tile_fits(sb, jrx, a0w, bminw);    // uniform over the wave
    const bool mine = have_tile && sb.amax >= sb.amin && fit != SLOW;   // this launch's share of the (tile, column) pairs
    // this key's coordinates inside the chunk for this column
    auto chunk_coords = [&](int kb, float& tcol, float& trow) {
      const TapRec rk = stash[16 * kb + li];
      const float tx = jrx + rk.b;
      const float xf = floorf(tx);
      tcol = rk.ys < -50.0f ? -8.0f : (xf - (float)x0) + (tx - xf);
      trow = rk.a - (float)a0;
    };
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (g >= 2) {
        u32x4 cw = {0u, 0u, 0u, 0u};
        if (!SLOW && mine) {
          float tcol, trow;
          chunk_coords(kb, tcol, trow);
          cw = __builtin_bit_cast(u32x4, cell_weights<PREC>(tcol, trow, g - 2).v);
        }
        bk[kb] = __builtin_bit_cast(bf16x8, cw);
      }
      zc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    win_b = win_base + (j % 3) * win_bytes;
    {
      const int iq = li + da, pq = iq & 1;                       // plain reads: window row of BEV row li (row block 0)
      wq0 = ((pq * NCW + dx + 2 * max(g - 2, 0)) * NRWD + ((iq - pq) >> 1)) * 4;
      wq1 = wq0 + NRWD * 4;
      const int it = 4 * g + cell_r + da, pt = it & 1;           // transposed reads: rows 4 g + cell_r .. + 3 and + 16 ..
      wth = ((pt * NCW + dx + cell_c) * NRWD + ((it - pt) >> 1)) * 4;
      wtl = wth + 2 * NCW * NRWD * 4;
    }
    for (int st = 0; st < nstep; ++st, ++e) {
      __syncthreads();
      if (!mine) continue;
#pragma unroll
      for (int u = 0; u < SPB; ++u)
        if (SPB * st + u < nslab)
          slab(std::integral_constant<bool, !SLOW>{}, smem + (e & 1) * L::BUF + u * L::SLAB, 32 * (SPB * st + u));
    }
    // ---- the column's bias-position gradients out of Z: this lane holds chunk column c = g, rows 0..3 of its key ----
    if (!SLOW && mine) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float tc, tr;
        chunk_coords(kb, tc, tr);
        const float c0 = floorf(tc), r0 = floorf(tr);
        const float wxg = hat((float)g - tc);
        const float z0 = zc[kb][0], z1 = zc[kb][1], z2 = zc[kb][2], z3 = zc[kb][3];
        // rows r0, r0 + 1 (r0 in 0..2 for a key inside the chunk)
        const float zl = r0 < 0.5f ? z0 : (r0 < 1.5f ? z1 : z2);
        const float zh = r0 < 0.5f ? z1 : (r0 < 1.5f ? z2 : z3);
        const float zrow = hat(0.f - tr) * z0 + hat(1.f - tr) * z1 + hat(2.f - tr) * z2 + hat(3.f - tr) * z3;
        const float sg = ((float)g == c0 + 1.0f ? 1.0f : 0.f) - ((float)g == c0 ? 1.0f : 0.f);
        acc_a[kb] += wxg * (zh - zl);      // summed over the four lane groups at the end (the sum is linear)
        acc_b[kb] += sg * zrow;
      }
    }
  }
  __syncthreads();   // the producer's closing barrier

  if (!have_tile) return;
  // ---- per key: tap-weight gradients -> sampling-position gradient; everything summed over the four lane groups ----
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const TapRec rk = stash[16 * kb + li];
    const float ys = rk.ys, xs = rk.xs;
    const float y0 = floorf(ys), x0f = floorf(xs);
    float gy = 0.f, gx = 0.f;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int t = 4 * g + e2;                       // slot; taps are t < 12: (r, c) = (t / 3, t % 3)
      const float r = (float)(t / 3), c = (float)(t % 3);
      const float dwy = (r == y0 + 1.0f ? 1.0f : 0.f) - (r == y0 ? 1.0f : 0.f);
      const float dwx = (c == x0f + 1.0f ? 1.0f : 0.f) - (c == x0f ? 1.0f : 0.f);
      const float z = g < 3 ? zt[kb][e2] + BEVR_LOG2E * zv[kb][e2] : 0.f;
      gy += z * dwy * hat(c - xs);
      gx += z * hat(r - ys) * dwx;
    }
    float va = acc_a[kb], vb = acc_b[kb];
    gy += __shfl_xor(gy, 16); gx += __shfl_xor(gx, 16); va += __shfl_xor(va, 16); vb += __shfl_xor(vb, 16);
    gy += __shfl_xor(gy, 32); gx += __shfl_xor(gx, 32); va += __shfl_xor(va, 32); vb += __shfl_xor(vb, 32);
    const int n = tile * 32 + 16 * kb + li;
    if (g == 0 && n < d.N) {
      const size_t idx = (size_t)prob * d.Np + n;
      atomicAdd(dkey_a + idx, va);
      atomicAdd(dkey_b + idx, vb);
      atomicAdd(dkey_y + idx, gy);
      atomicAdd(dkey_x + idx, gx);
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* G, const void* H, const void* tap_ws, const float* table_t, float* dkey_a,
           float* dkey_b, float* dkey_y, float* dkey_x, hipStream_t st) {
  typedef LdsK L;
  const int n_ph = d.n_prob * d.heads;
  const int n_tiles = d.Np / 32;
  const int n_wg_ph = (n_tiles + NKW - 1) / NKW;
  const size_t lds = 2 * L::BUF + (size_t)3 * 4 * NCW * win_dwords(d.Sp) * 4 + NKW * 32 * sizeof(TapRec);
  if (lds > 160 * 1024) return BEVR_E_SHAPE;
  const long long grid = (long long)((n_ph + 7) / 8) * 8 * n_wg_ph;
  if (grid > 0x7fffffffLL) return BEVR_E_SHAPE;
  hipLaunchKernelGGL((attn_tap_bwd_k_kernel<PREC, false>), dim3((unsigned)grid), dim3(64 * (NKW + 1)), lds, st, d, (const char*)G,
                     (const char*)H, (const char*)tap_ws, table_t, dkey_a, dkey_b, dkey_y, dkey_x, n_wg_ph);
  const int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL((attn_tap_bwd_k_kernel<PREC, true>), dim3((unsigned)grid), dim3(64 * (NKW + 1)), lds, st, d, (const char*)G,
                     (const char*)H, (const char*)tap_ws, table_t, dkey_a, dkey_b, dkey_y, dkey_x, n_wg_ph);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_tap_bwd_k(const bevr_attn_desc* d, const void* G, const void* H, const void* tap_ws,
                                   const float* table_t, float* dkey_a, float* dkey_b, float* dkey_y, float* dkey_x,
                                   void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!G || !H || !tap_ws || !table_t || !dkey_a || !dkey_b || !dkey_y || !dkey_x) return BEVR_E_NULL;
  if (d->groups != 1) return BEVR_E_SHAPE;
  if (!bevr_aligned16(G) || !bevr_aligned16(H) || !bevr_aligned16(tap_ws)) return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16) return launch<BEVR_PREC_BF16>(*d, G, H, tap_ws, table_t, dkey_a, dkey_b, dkey_y, dkey_x, st);
  if (d->precision == BEVR_PREC_F16) return launch<BEVR_PREC_F16>(*d, G, H, tap_ws, table_t, dkey_a, dkey_b, dkey_y, dkey_x, st);
  return BEVR_E_PRECISION;
}
