// Attention backward, key side, over a CELL-SORTED key segment (attn_cell.h): dK, dV and the gradients of the keys'
// table coordinates (a_n, b_n).  Counterpart of attn_bwd_k.hip: key-stationary ("key on the lane", tiles S[query][key]),
// a workgroup owns 384 consecutive keys (12 waves x 32) and sweeps all query tiles, so every per-key sum stays in
// registers and dK, dV need no reduction across workgroups.
//
// For a wave whose 32 keys fit one table chunk (attn_cell.h) the bias of a tile is one MFMA,
//   bias[i][n] = sum_k' Tsh[k'][i] W[n][k'],
// with W (lane = key) built once per BEV column j and reused over the column's row blocks, and the derivatives of
// the bias with respect to the key's table coordinates are two more MFMAs against the derivative weights
//   d bias / d a = sum_k' Tsh[k'][i] wx[c] dy[r],   dy = -1 at the key's first tap row, +1 at the second
//   d bias / d b = sum_k' Tsh[k'][i] dx[c] wy[r],   dx likewise over columns
// (what F.grid_sample's backward computes: model/SCA_deform_attn.py:379-389 of the reference), contracted with dS per
// lane.  The table operand changes with every tile (its rows shift with the row block, its columns with j): four
// 8-byte loads per lane from the pair table, shared through L1 / L2 by the waves of the workgroup (cell-sorted keys:
// neighbouring tiles sit in the same cells).  (Wave, column) combinations whose keys do not fit one chunk are left to a
// second, SLOW pass of the same kernel (per-pair gather from the table in global memory, any key set; it ADDS its dK, dV to
// the fast pass's, and its workgroups exit at once when none of their waves has such a column).
#include <type_traits>
#include "attn_cell.h"
#include "attn_kstage.h"


namespace {

// 16-bit modes: 12 waves x 32 keys, 3 waves per SIMD (168 registers): one workgroup per CU.  f32-layout modes: fragments
// twice as wide -- 8 waves, 2 per SIMD (256 registers), no spills.
template <int PREC> constexpr int twc() { return is16(PREC) ? 768 : 512; }
template <int PREC> constexpr int keys_wgc() { return twc<PREC>() / 2; }

// derivative weights of this lane's key over the chunk (see the header): cn / rn = first tap column / row relative to the
// chunk origin (integers), wx / wy as cell_weights
template <int PREC, std::enable_if_t<is16(PREC), int> = 0>
__device__ __forceinline__ void cell_dweights(float tcol, float trow, int h, CellFrag<PREC>& wyf, CellFrag<PREC>& wxf) {
  const float cn = floorf(tcol), rn = floorf(trow);
  float wx[2], dx[2], wy[4], dy[4];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const float c = (float)(2 * h + cc);
    wx[cc] = hat(c - tcol);
    dx[cc] = c == cn ? -1.f : (c == cn + 1.f ? 1.f : 0.f);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    wy[r] = hat((float)r - trow);
    dy[r] = (float)r == rn ? -1.f : ((float)r == rn + 1.f ? 1.f : 0.f);
  }
  u32x4 a, b;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    a[2 * cc] = Half<PREC>::pack2(wx[cc] * dy[0], wx[cc] * dy[1]);
    a[2 * cc + 1] = Half<PREC>::pack2(wx[cc] * dy[2], wx[cc] * dy[3]);
    b[2 * cc] = Half<PREC>::pack2(dx[cc] * wy[0], dx[cc] * wy[1]);
    b[2 * cc + 1] = Half<PREC>::pack2(dx[cc] * wy[2], dx[cc] * wy[3]);
  }
  wyf.v = __builtin_bit_cast(bf16x8, a);
  wxf.v = __builtin_bit_cast(bf16x8, b);
}
template <int PREC, std::enable_if_t<!is16(PREC), int> = 0>
__device__ __forceinline__ void cell_dweights(float tcol, float trow, int h, CellFrag<PREC>& wyf, CellFrag<PREC>& wxf) {
  const float cn = floorf(tcol), rn = floorf(trow);
  float wy[2], dy[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const float r = (float)(2 * rr + h);
    wy[rr] = hat(r - trow);
    dy[rr] = r == rn ? -1.f : (r == rn + 1.f ? 1.f : 0.f);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float wx = hat((float)c - tcol);
    const float dx = (float)c == cn ? -1.f : ((float)c == cn + 1.f ? 1.f : 0.f);
    wyf.v[2 * c] = wx * dy[0];
    wyf.v[2 * c + 1] = wx * dy[1];
    wxf.v[2 * c] = dx * wy[0];
    wxf.v[2 * c + 1] = dx * wy[1];
  }
  cell_split(wyf);
  cell_split(wxf);
}

template <int PREC, bool SLOW>
__global__ __launch_bounds__(twc<PREC>(), is16(PREC) ? 3 : 2) void attn_cell_bwd_k_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ Qt, const char* __restrict__ K,
    const char* __restrict__ V, const char* __restrict__ key_ws, const char* __restrict__ table_pair,
    const char* __restrict__ dO, const char* __restrict__ dOt, const float* __restrict__ LSE,
    const float* __restrict__ delta, const float* __restrict__ grad_scale, float* __restrict__ dK,
    float* __restrict__ dV, float* __restrict__ dkey_a, float* __restrict__ dkey_b) {
  typedef LdsK<PREC, 1> L;
  constexpr int TWC = twc<PREC>(), KEYS_WGC = keys_wgc<PREC>();
  // fp16 mode (include/bevrender_hip.h, grad_scale[2..5]): P' = P 2^kp, dS16 = P' (dP - delta) c2
  const float kp16 = PREC == BEVR_PREC_F16 ? grad_scale[2] : 0.f, c2_16 = PREC == BEVR_PREC_F16 ? grad_scale[3] : 1.f;
  const float ds_inv = PREC == BEVR_PREC_F16 ? grad_scale[4] : 1.f, p_inv = PREC == BEVR_PREC_F16 ? grad_scale[5] : 1.f;
  constexpr int EB = L::EB;
  __shared__ __attribute__((aligned(16))) char smem[2 * L::BUF];

  const int n_kb = (d.Np + KEYS_WGC - 1) / KEYS_WGC;
  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / n_kb) * 8 + xcd;
  if (ph >= n_ph) return;
  const int kblk = slot % n_kb;
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int grp = hd / (d.heads / d.groups);
  const int qb = prob / d.q_div;

  const int tid = threadIdx.x, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Mp = d.S * d.Sp;
  const int n_rb = d.Sp / 32;

  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = V + ((size_t)ph * d.Np) * 32 * EB;
  const int pg = prob * d.groups + grp;
  const KeyW* kws = reinterpret_cast<const KeyW*>(key_ws) + (size_t)pg * d.Np;
  const StepBox* kbox = reinterpret_cast<const StepBox*>(key_ws + key_ws_box_offset(d)) + (size_t)pg * (d.Np / 32);
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const int Hp8 = d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));

  // ---- this wave's 32 keys -----------------------------------------------------------------------------------
  const int k0 = kblk * KEYS_WGC + wave * 32;
  const bool wave_live = k0 < d.Np;                     // wave-uniform
  const int key = wave_live ? k0 + lq : lq;             // dead waves read a valid address, never store
  const bool dead = !wave_live || key >= d.N;           // padded key: P = 0
  const KeyW kw = kws[key];
  const StepBox sb = kbox[wave_live ? k0 / 32 : 0];     // the tile's box (scalar)
  const bool tile_live = wave_live && sb.amax >= sb.amin;
  const bool any_dead = __any(dead);
  // SLOW: the BEV columns in which SOME wave's tile does not fit a chunk, listed in ascending order (deterministic);
  // the sweep below visits only those (a workgroup with one slow (tile, column) pair used to sweep all S columns)
  __shared__ unsigned char col_flag[SLOW ? 1024 : 1];
  __shared__ short col_list[SLOW ? 1024 : 1];
  __shared__ int col_count;
  if constexpr (SLOW) {
    for (int jj = tid; jj < d.S; jj += TWC) col_flag[jj] = 0;
    __syncthreads();
    if (tile_live)
      for (int jj = lane; jj < d.S; jj += 64)
        if (!make_celltile(sb, (float)jj * rx).fast) col_flag[jj] = 1;   // benign race: every writer stores 1
    __syncthreads();
    if (tid < 64) {
      int cnt = 0;
      for (int b0 = 0; b0 < d.S; b0 += 64) {
        const bool f = b0 + lane < d.S && col_flag[b0 + lane];
        const unsigned long long mask = __ballot(f);
        if (f) col_list[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (short)(b0 + lane);
        cnt += __popcll(mask);
      }
      if (lane == 0) col_count = cnt;
    }
    __syncthreads();
    if (col_count == 0) return;
  }
  Frag<PREC> kf, vf;
  kf.load(Kh + (size_t)key * 32 * EB, hi);
  vf.load(Vh + (size_t)key * 32 * EB, hi);
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
  float da = 0.f, db = 0.f;

  // per BEV column: the chunk, this key's weights over it, and the chunk's table columns (clamped into the padded table
  // once per column; per tile only the row offset changes)
  CellTile ct = make_celltile(sb, 0.f);
  CellFrag<PREC> wf, wyf, wxf;
  float jr = 0.f;
  constexpr int NCOLP = is16(PREC) ? 2 : 4;   // table columns this lane half reads (attn_cell.h lane maps)
  const char* colp[NCOLP];
#pragma unroll
  for (int c = 0; c < NCOLP; ++c) colp[c] = tbl;
  // raw table entries of a tile's chunk for this lane: bf16 mode 2 columns x (T[y..y+1], T[y+2..y+3]); f32 mode 4 columns x
  // (T[y + hi], T[y + hi + 2])
  constexpr int NRAW = is16(PREC) ? 4 : 8;
  typedef typename std::conditional<is16(PREC), f32x2, float>::type raw_t;
  raw_t traw[NRAW];
#pragma unroll
  for (int k = 0; k < NRAW; ++k) traw[k] = raw_t{};
  auto load_raw = [&](raw_t* dst, int rb_) {
    // rows ct.a0 + rb * 32 + lq (+ 0..3): never below the padded table's first row (a0 >= -(Sp + 1), y_off = Sp + 2)
    const int yr = ct.a0 + rb_ * 32 + lq + d.y_off;
    if constexpr (is16(PREC)) {
      const int e0 = min(yr, d.Hp - 1) * 8, e2 = min(yr + 2, d.Hp - 1) * 8;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        dst[2 * c] = *reinterpret_cast<const f32x2*>(colp[c] + e0);
        dst[2 * c + 1] = *reinterpret_cast<const f32x2*>(colp[c] + e2);
      }
    } else {
      const int ea = min(yr + hi, d.Hp - 1) * 8, eb = min(yr + hi + 2, d.Hp - 1) * 8;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dst[2 * c] = *reinterpret_cast<const float*>(colp[c] + ea);
        dst[2 * c + 1] = *reinterpret_cast<const float*>(colp[c] + eb);
      }
    }
  };

  QStage<PREC, TWC, 1> qs;
  qs.init(tid, Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB, dO + ((size_t)ph * Mp) * 32 * EB,
          Qt + ((size_t)(qb * d.heads + hd) * 32) * Mp * EB, dOt + ((size_t)ph * 32) * Mp * EB,
          LSE + (size_t)ph * Mp, delta + (size_t)ph * Mp, Mp, kp16);
  // tile it = (BEV column j = it / n_rb, row block rb = it % n_rb) is the 32 packed queries [32 it, 32 it + 32)
  const int n_col = SLOW ? col_count : d.S;
  const int n_it = n_col * n_rb;
  // iteration -> (BEV column, row block) -> first packed query of the tile
  auto col_of = [&](int it_) { const int cj = it_ / n_rb; return SLOW ? (int)col_list[cj] : cj; };
  auto mq_of = [&](int it_) { return (size_t)(col_of(it_) * n_rb + it_ % n_rb) * 32; };
  qs.load(tid, mq_of(0));
  qs.store(tid, smem);
  __syncthreads();

  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    const char* base = smem + buf * L::BUF;
    if (it + 1 < n_it) qs.load(tid, mq_of(it + 1));
    const int j = col_of(it), rb = it % n_rb;

    if (rb == 0) {
      jr = (float)j * rx;
      ct = make_celltile(sb, jr);
    }
    if (tile_live && (bool)ct.fast != SLOW) {   // this pass's (wave, column) combinations (wave-uniform)
      if (rb == 0) {   // new BEV column: chunk origin and weights of this wave's keys
        if constexpr (!SLOW) {
          float tcol, trow;
          cell_coords(kw, jr, ct.x0, dead, tcol, trow);
          wf = cell_weights<PREC>(tcol, trow, hi);
          cell_dweights(tcol, trow, hi, wyf, wxf);
#pragma unroll
          for (int c = 0; c < NCOLP; ++c) {
            const int xc = ct.x0 + (is16(PREC) ? 2 * hi + c : c) + d.x_off;
            colp[c] = tbl + (size_t)max(0, min(xc, d.Wp - 1)) * Hp8;
          }
        }
      }
      // table operand of this (column, row block), lane = BEV row rb * 32 + lq: requested first, consumed after the
      // S and dP products (global loads served by L1 / L2: the waves of the workgroup sit in the same cells)
      // table operand of this (column, row block), lane = BEV row rb * 32 + lq (global loads served by L1 / L2: the waves of
      // the workgroup sit in the same cells).  Its raw entries live in `traw`; the NEXT row block's are requested into the
      // same registers at the END of this tile (below), so only a column's first row block waits for its
      // loads.  (A second set of registers for the prefetch: 11 spills, 28.5 -> 36.6 ms.  Without the table loads: -15 %.)
      if constexpr (!SLOW) {
        if (rb == 0) load_raw(traw, 0);   // later row blocks: requested during the previous tile, see below
      }
      const f32x4* rc = reinterpret_cast<const f32x4*>(base + 2 * L::TILE_Q + 2 * L::TILE_T);
      f32x16 s, dp;
      {
        Frag<PREC> qf;
        qf.load(base + lq * L::STRIDE, hi);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {   // rows 8 g4 + 4 hi + 0..3
          const f32x4 l4 = rc[2 * g4 + hi];
#pragma unroll
          for (int k = 0; k < 4; ++k) s[4 * g4 + k] = l4[k];
        }
        s = mma_frag(qf, kf, s);      // S[q][key] - LSE[q]
      }
      {
        Frag<PREC> dof;
        dof.load(base + L::TILE_Q + lq * L::STRIDE, hi);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 d4 = rc[8 + 2 * g4 + hi];
#pragma unroll
          for (int k = 0; k < 4; ++k) dp[4 * g4 + k] = d4[k];
        }
        dp = mma_frag(dof, vf, dp);   // dP[q][key] - delta[q]
      }
      float sa = 0.f, sbb = 0.f;
      if constexpr (!SLOW) {
        CellFrag<PREC> tf;
        if constexpr (is16(PREC)) {
          u32x4 w;
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = Half<PREC>::pack2(traw[k][0], traw[k][1]);
          tf.v = __builtin_bit_cast(bf16x8, w);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) tf.v[k] = traw[k];
          cell_split(tf);
        }
        s = mma_cell(tf, wf, s);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 pp = {fast_exp2(s[r]), fast_exp2(s[r + 1])};
          if (any_dead) pp = dead ? f32x2{0.f, 0.f} : pp;
          f32x2 ds = pp * f32x2{dp[r], dp[r + 1]};   // dS = P (dP - delta); ln2 folded into the epilogue
          if constexpr (PREC == BEVR_PREC_F16) ds *= f32x2{c2_16, c2_16};
          s[r] = pp[0]; s[r + 1] = pp[1];
          dp[r] = ds[0]; dp[r + 1] = ds[1];
        }
        // the two derivative products one after the other: they share 16 registers
        {
          f32x16 g;
#pragma unroll
          for (int r = 0; r < 16; ++r) g[r] = 0.f;
          g = mma_cell(tf, wyf, g);   // d bias / d a  [q][key]
#pragma unroll
          for (int r = 0; r < 16; ++r) sa = fmaf(dp[r], g[r], sa);
        }
        {
          f32x16 g;
#pragma unroll
          for (int r = 0; r < 16; ++r) g[r] = 0.f;
          g = mma_cell(tf, wxf, g);   // d bias / d b
#pragma unroll
          for (int r = 0; r < 16; ++r) sbb = fmaf(dp[r], g[r], sbb);
        }
      } else {
        // per-pair gather from the table in global memory
        const float wy0 = 1.0f - kw.fy;
        const float tx = jr + kw.b;
        const float xf = floorf(tx);
        const float fx = tx - xf;
        const char* tp = tbl + (unsigned)((int)xf * Hp8 + kw.aoff + (rb * 32 + 4 * hi) * 8);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x2 t0 = *reinterpret_cast<const f32x2*>(tp + crow(r, 0) * 8);
          const f32x2 t1 = *reinterpret_cast<const f32x2*>(tp + Hp8 + crow(r, 0) * 8);
          const float u0 = t0[0] * wy0 + t0[1] * kw.fy;
          const float u1 = t1[0] * wy0 + t1[1] * kw.fy;
          const float sv = s[r] + u0 + fx * (u1 - u0);
          const float p = dead ? 0.f : fast_exp2(sv);
          const float ds = p * dp[r] * c2_16;
          s[r] = p;
          dp[r] = ds;
          const float ga = (t0[1] - t0[0]) + fx * ((t1[1] - t1[0]) - (t0[1] - t0[0]));
          sa = fmaf(ds, ga, sa);
          sbb = fmaf(ds, u1 - u0, sbb);
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight (rare path, register budget)
        }
      }
      da += sa;
      db += sbb;
      {
        Frag<PREC> dotf;
        load_perm(dotf, base + 2 * L::TILE_Q + L::TILE_T + lq * L::TSTRIDE, hi);
        dv = mma_acc_b(dotf, s, dv);
      }
      {
        Frag<PREC> qtf;
        load_perm(qtf, base + 2 * L::TILE_Q + lq * L::TSTRIDE, hi);
        dk = mma_acc_b(qtf, dp, dk);
      }
      // the next row block's table entries, requested at the tile's end (register pressure is past its peak) and consumed
      // in the middle of the next tile
      if constexpr (!SLOW) {
        if (rb + 1 < n_rb) load_raw(traw, rb + 1);
      }
    }

    if (it + 1 < n_it) qs.store(tid, smem + (buf ^ 1) * L::BUF);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------
  if (wave_live) {
    float* kr = dK + ((size_t)ph * d.Np + key) * 32;
    float* vr = dV + ((size_t)ph * d.Np + key) * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
      if constexpr (SLOW) {   // this wave owns the rows: plain read-modify-write after the fast pass
        a = *reinterpret_cast<const f32x4*>(kr + 8 * g4 + 4 * hi);
        b = *reinterpret_cast<const f32x4*>(vr + 8 * g4 + 4 * hi);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { a[k] += BEVR_LN2 * ds_inv * dk[4 * g4 + k]; b[k] += p_inv * dv[4 * g4 + k]; }
      *reinterpret_cast<f32x4*>(kr + 8 * g4 + 4 * hi) = a;
      *reinterpret_cast<f32x4*>(vr + 8 * g4 + 4 * hi) = b;
    }
    const float sa = BEVR_LN2 * ds_inv * (da + __shfl_xor(da, 32));
    const float sb2 = BEVR_LN2 * ds_inv * (db + __shfl_xor(db, 32));
    if (hi == 0) {
      atomicAdd(dkey_a + (size_t)pg * d.Np + key, sa);
      atomicAdd(dkey_b + (size_t)pg * d.Np + key, sb2);
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* Qt, const void* K, const void* V, const void* key_ws,
           const float* table_pair, const void* dO, const void* dOt, const float* LSE, const float* delta,
           const float* gs, float* dK, float* dV, float* dka, float* dkb, hipStream_t st) {
  constexpr int TWC = twc<PREC>(), KEYS_WGC = keys_wgc<PREC>();
  const int n_kb = (d.Np + KEYS_WGC - 1) / KEYS_WGC;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_kb;
  hipLaunchKernelGGL((attn_cell_bwd_k_kernel<PREC, false>), dim3(grid), dim3(TWC), 0, st, d, (const char*)Q,
                     (const char*)Qt, (const char*)K, (const char*)V, (const char*)key_ws, (const char*)table_pair,
                     (const char*)dO, (const char*)dOt, LSE, delta, gs, dK, dV, dka, dkb);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL((attn_cell_bwd_k_kernel<PREC, true>), dim3(grid), dim3(TWC), 0, st, d, (const char*)Q,
                     (const char*)Qt, (const char*)K, (const char*)V, (const char*)key_ws, (const char*)table_pair,
                     (const char*)dO, (const char*)dOt, LSE, delta, gs, dK, dV, dka, dkb);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_cell_bwd_k(const bevr_attn_desc* d, const void* Q, const void* Qt, const void* K, const void* V,
                                    const void* key_ws, const float* table_pair, const void* dO, const void* dOt,
                                    const float* LSE, const float* delta, const float* grad_scale, float* dK,
                                    float* dV, float* dkey_a, float* dkey_b, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (d->S > 1024) return BEVR_E_SHAPE;   // the slow pass lists its BEV columns in LDS (1024 entries)
  if (!Q || !Qt || !K || !V || !key_ws || !table_pair || !dO || !dOt || !LSE || !delta || !dK || !dV || !dkey_a ||
      !dkey_b || (d->precision == BEVR_PREC_F16 && !grad_scale))
    return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(Qt) || !bevr_aligned16(K) || !bevr_aligned16(V) || !bevr_aligned16(dO) ||
      !bevr_aligned16(dOt) || !bevr_aligned16(dK) || !bevr_aligned16(dV) || !bevr_aligned16(table_pair) ||
      !bevr_aligned16(key_ws))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16>(*d, Q, Qt, K, V, key_ws, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV, dkey_a,
                                  dkey_b, st);
  if (d->precision == BEVR_PREC_F16)
    return launch<BEVR_PREC_F16>(*d, Q, Qt, K, V, key_ws, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV, dkey_a,
                                 dkey_b, st);
  if (d->precision == BEVR_PREC_BF16X3)
    return launch<BEVR_PREC_BF16X3>(*d, Q, Qt, K, V, key_ws, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV, dkey_a,
                               dkey_b, st);
  return launch<BEVR_PREC_F32>(*d, Q, Qt, K, V, key_ws, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV, dkey_a,
                               dkey_b, st);
}
