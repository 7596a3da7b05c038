// Attention backward, query side: dQ and the rpe-table gradient.
// Same orientation, LDS table region and XCD mapping as the forward (attn_fwd.hip): S^T[key][query] tiles with
// the query on the lane, so
//   * P^T = exp2(S^T - LSE[q]) and dS^T = ln2 * P^T * (dP^T - delta[q]) need only lane-local constants
//     (preloaded into the accumulators of the two MFMA chains),
//   * dQ^T[c][q] += K^T[c][key] dS^T[key][q] takes dS^T straight from the accumulator (B operand),
//   * the table gradient of a (query tile x key step) block lands inside the same box the bias was read
//     from: each wave accumulates it in a private LDS window (plain read-modify-write; 32 lanes = 32
//     consecutive rows: conflict-free) that is flushed to HBM with contiguous float atomics only when the
//     region moves.  Steps whose box does not fit scatter straight to global memory.
// Workgroup = 8 waves (2 per SIMD: the read-modify-write chains are LDS-latency bound) on one 32-row x
// 8-column query tile: wave w owns column pair (w & 3) and the key half (w >> 2) of every 64-key step; the two
// key halves' dQ partial sums are merged through LDS at the end.
// Recomputes S from Q, K and the bias instead of storing any (M x N) tensor.
// Gradient semantics: see include/bevrender_hip.h (log2-domain inputs as handed in).
#include "attn_tile.h"

namespace {

constexpr int TQ = 512;   // threads per workgroup
constexpr int NWAVE = TQ / 64;
constexpr int NQ = 2;     // query columns per wave

template <int PREC> struct LdsQ {
  static constexpr int EB = Elem<PREC>::bytes;
  // region capacity (columns of the shared table window) and per-wave accumulation-window width
  static constexpr int CAP_MAX = PREC == BEVR_PREC_BF16 ? 64 : 48;
  static constexpr int WACC = PREC == BEVR_PREC_BF16 ? 40 : 32;
  static constexpr int R_STRIDE = 32 * EB + 16;   // row-layout tiles (K, V): bytes per key row
  static constexpr int T_STRIDE = KT * EB + 16;   // transposed tile (Kt): bytes per channel row
  static constexpr int R_BYTES = KT * R_STRIDE;
  static constexpr int T_BYTES = 32 * T_STRIDE;
  static constexpr int C_BYTES = KT * 16 + 32;
  static constexpr int BUF = 2 * R_BYTES + T_BYTES + C_BYTES;
  static constexpr int WIN = CAP_MAX * WIN_PITCH * 8;
  static constexpr int ACC1 = (WACC + 3) * WIN_PITCH * 4;   // one wave's window (floats, WIN_PITCH rows per column) + dummy columns (a dummy update touches [off, off + AP + 1])
  static constexpr int TOTAL = 2 * BUF + WIN + NWAVE * ACC1;
};

template <int PREC>
__global__ __launch_bounds__(TQ) void attn_bwd_q_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ Kt,
    const char* __restrict__ V, const float* __restrict__ key_a, const float* __restrict__ key_b,
    const char* __restrict__ table_pair, const char* __restrict__ dO, const float* __restrict__ LSE,
    const float* __restrict__ delta, float* __restrict__ dQ, float* __restrict__ dtable) {
  typedef LdsQ<PREC> L;
  constexpr int EB = L::EB;
  constexpr int WACC = L::WACC;
  constexpr int AP = WIN_PITCH;   // accumulation-window row pitch (floats)
  static_assert(L::TOTAL <= 160 * 1024, "LDS budget");
  // Three separate LDS objects, not one carved buffer: the compiler then knows that the read-modify-write
  // stores into the accumulation windows cannot alias the staged tiles, key constants or table window, and
  // keeps hoisting those loads across them (with one buffer every load waited behind the previous key's
  // stores and the loop ran at LDS latency: ~900 cycles per 64 pairs).
  __shared__ __attribute__((aligned(16))) char smem[2 * L::BUF];
  __shared__ __attribute__((aligned(16))) char win[L::WIN];
  __shared__ __attribute__((aligned(16))) float acc_all[NWAVE * (WACC + 3) * AP];

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
  const int cp = wave & 3, kh = wave >> 2;
  float* acc = acc_all + wave * ((WACC + 3) * AP);
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
  // A wave's two columns must never share table cells (their updates go out in one LDS instruction): adjacent
  // columns when rx >= 2 (SCA: tx advances ~5 cells per column), columns 4 apart otherwise (TSA: rx = 1).
  const bool rx2 = rx >= 2.0f;
  const int dj = rx2 ? 1 : 4;
  // Region capacity: chosen so that a step box inside the region implies every wave's two columns stay inside
  // its WACC-wide accumulation window: the wave spans dj of the block's columns, the region all of them.
  const int cap = min(L::CAP_MAX, WACC - 4 + (int)floorf((float)(j_last - j_first - dj) * rx));

  Frag<PREC> qf[NQ], dof[NQ];
  float jrx[NQ], lse[NQ], dlt[NQ];
  int jcol[NQ];
  bool live[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    int j = j_first + (rx2 ? cp * NQ + t : cp + 4 * t);
    jcol[t] = j;
    live[t] = j < d.S;
    int jc = live[t] ? j : d.S - 1;
    jrx[t] = (float)jc * rx;
    size_t mq = (size_t)jc * d.Sp + i0 + lq;
    qf[t].load(Qh + mq * 32 * EB, hi);
    dof[t].load(dOh + mq * 32 * EB, hi);
    lse[t] = LSE[(size_t)ph * Mp + mq];
    dlt[t] = delta[(size_t)ph * Mp + mq];
    if (!live[t]) {  // duplicate of column S-1: must contribute nothing
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
  // first window column of this wave, relative to the region's column 0 (lower bound of floor(tx) - ax0)
  const int wcol0 = max(0, (int)floorf(jrx[0] - jrx_lo) - 1);
  const int ilane = i0 + lq;
  const int rowoff = ilane * 8;
  const int xoffHp = d.x_off * d.Hp;
  const int dummy_off = WACC * AP + lane;   // per-lane cell of the dummy column: target of non-live columns

  f32x16 dq[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  constexpr int RCH_ROW = 32 * EB / 16;
  constexpr int TCH_ROW = KT * EB / 16;
  constexpr int CH = KT * RCH_ROW;            // 16-B chunks per tile: 256 (bf16) / 512 (f32)
  static_assert(CH <= TQ, "one chunk per thread per tile");
  u32x4 stK, stV, stT;
  float st_a = 0.f, st_b = 0.f;
  const int n_step = d.Np / KT;

  auto stage_load = [&](int step) {
    if (tid < CH) {
      stK = *reinterpret_cast<const u32x4*>(Kh + ((size_t)step * CH + tid) * 16);
      stV = *reinterpret_cast<const u32x4*>(Vh + ((size_t)step * CH + tid) * 16);
      int tr = tid / TCH_ROW, tc = tid % TCH_ROW;
      stT = *reinterpret_cast<const u32x4*>(Kth + ((size_t)tr * d.Np + (size_t)step * KT) * EB + tc * 16);
    }
    if (tid < KT) { st_a = ka[step * KT + tid]; st_b = kb[step * KT + tid]; }
  };
  auto stage_store = [&](int buf, int step) {
    char* base = smem + buf * L::BUF;
    if (tid < CH) {
      int ro = (tid / RCH_ROW) * L::R_STRIDE + (tid % RCH_ROW) * 16;
      *reinterpret_cast<u32x4*>(base + ro) = stK;
      *reinterpret_cast<u32x4*>(base + L::R_BYTES + ro) = stV;
      *reinterpret_cast<u32x4*>(base + 2 * L::R_BYTES + (tid / TCH_ROW) * L::T_STRIDE + (tid % TCH_ROW) * 16) = stT;
    }
    if (tid < KT) {   // exactly wave 0
      WinInfo wi;
      KeyW kw = stage_keys(st_a, st_b, step * KT + tid < d.N, d, jrx_lo, jrx_hi, cap, wi);
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
  bool acc_live = false;   // the accumulation windows hold un-flushed gradient

  // flush this wave's window of region `r`: per column one contiguous run of 64 floats, non-zeros only
  auto flush = [&](const Region& r) {
    const size_t y0 = (size_t)(i0 + r.ay0 + d.y_off) + lane;
    for (int c = 0; c < WACC; ++c) {
      float v = acc[c * AP + lane];
      if (v != 0.f) atomicAdd(dtb + (size_t)(r.ax0 + wcol0 + c + d.x_off) * Hq + y0, v);
    }
  };

  for (int step = 0; step < n_step; ++step) {
    const int buf = step & 1;
    const char* base = smem + buf * L::BUF;
    if (step + 1 < n_step) stage_load(step + 1);
    const WinInfo wi = *reinterpret_cast<const WinInfo*>(base + 2 * L::R_BYTES + L::T_BYTES + KT * 16);
    const bool use_win = wi.ok != 0;   // workgroup-uniform
    if (use_win && !region_contains(rg, wi, cap)) {
      // every wave finished the previous step (barrier at the end of the loop body): safe to drain and move
      if (acc_live) flush(rg);
      rg = region_anchor(wi, d, i0, cap);
      load_region(win, tbl, d, rg, i0, cap, NWAVE, wave, lane);
      for (int c = 0; c < WACC; ++c) acc[c * AP + lane] = 0.f;
      acc_live = true;
      __syncthreads();
    }
    const float ax0_f = (float)rg.ax0;
    const int drow = (wi.amin - rg.ay0) + lq;

    {
      Frag<PREC> kf, vkf, ktf;
      kf.load(base + (kh * 32 + lq) * L::R_STRIDE, hi);
      vkf.load(base + L::R_BYTES + (kh * 32 + lq) * L::R_STRIDE, hi);
      load_perm(ktf, base + 2 * L::R_BYTES + lq * L::T_STRIDE + kh * 32 * EB, hi);
      const KeyW* kc = reinterpret_cast<const KeyW*>(base + 2 * L::R_BYTES + L::T_BYTES) + kh * 32;
      const bool last = (step == n_step - 1) && d.N < d.Np;

      f32x16 s[NQ], dp[NQ];
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[t][r] = -lse[t]; dp[t][r] = -dlt[t]; }
        s[t] = mma_frag(kf, qf[t], s[t]);       // S^T - LSE
        dp[t] = mma_frag(vkf, dof[t], dp[t]);   // dP^T - delta
      }

      if (use_win) {
        // One basic block per key pair (no branches), so the scheduler overlaps the next keys' LDS reads with
        // this key's arithmetic.  Table-gradient accumulation, per accumulator register r (key A = crow(r,0) in
        // lanes 0-31, key B = crow(r,1) in lanes 32-63, for both of the wave's columns t):
        //   v_permlane32_swap regroups (t0: A|B, t1: A|B) into (A: t0|t1, B: t0|t1); then four plain
        //   read-modify-writes of this wave's private window, all 64 lanes active each:
        //     A own rows, A rows + 1, B own rows, B rows + 1.
        //   Inside one of them the 32 lanes of a half are 32 distinct rows and the two halves are the wave's two
        //   columns, which never share cells; A and B (which may: the projector pins every out-of-image key to
        //   one pixel) and the row / row + 1 taps are separated by program order -- a wave's LDS operations
        //   execute in order.  (LDS float atomics measured ~3.5x slower than this.)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const KeyW c = kc[crow(r, hi)];
          const float wy0 = 1.0f - c.fy;
          const float bl = c.b - ax0_f;
          const int row = (c.arow8 >> 3) + drow;
          const bool dead = last && step * KT + kh * 32 + crow(r, hi) >= d.N;
          int off[NQ];
          float c00[NQ], c01[NQ], c10[NQ], c11[NQ];
#pragma unroll
          for (int t = 0; t < NQ; ++t) {
            float tx = jrx[t] + bl;
            float xf = floorf(tx);
            float fx = tx - xf;
            const int xi = (int)xf;
            const char* p = win + xi * (WIN_PITCH * 8) + row * 8;
            f32x2 t0 = *reinterpret_cast<const f32x2*>(p);
            f32x2 t1 = *reinterpret_cast<const f32x2*>(p + WIN_PITCH * 8);
            float u0 = t0[0] * wy0 + t0[1] * c.fy;
            float u1 = t1[0] * wy0 + t1[1] * c.fy;
            float sv = s[t][r] + u0 + fx * (u1 - u0);
            if (dead) sv = BEVR_NEG_BIG;
            float ds = BEVR_LN2 * fast_exp2(sv) * dp[t][r];
            s[t][r] = ds;
            const float w0 = ds * (1.0f - fx), w1 = ds * fx;
            c00[t] = w0 * wy0; c01[t] = w0 * c.fy; c10[t] = w1 * wy0; c11[t] = w1 * c.fy;
            // window column of this wave; the clamp is a guard that by construction never binds
            const int xw = max(0, min(xi - wcol0, WACC - 2));
            off[t] = live[t] ? xw * AP + row : dummy_off;
          }
          // regroup by key: x[0] <- key A (t0 | t1), x[1] <- key B (t0 | t1)
          // v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second.  Written as inline
          // asm: with the builtin, hipcc (ROCm 7.2) treated the two results as equal after unrolling and applied
          // key A's update twice.  "s_nop 1" covers the VALU-write -> permlane-read hazard (2 wait states).
          auto swap32 = [](auto& a, auto& b) {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
          };
          swap32(off[0], off[1]);
          swap32(c00[0], c00[1]);
          swap32(c01[0], c01[1]);
          swap32(c10[0], c10[1]);
          swap32(c11[0], c11[1]);
#pragma unroll
          for (int k = 0; k < 2; ++k) {   // k = 0: key A, k = 1: key B
            float* g = acc + off[k];
            float v0 = g[0], v1 = g[AP];
            g[0] = v0 + c00[k];
            g[AP] = v1 + c10[k];
            // rows + 1: lane i's cell here is lane i+1's cell above, so these accesses must stay behind the
            // stores above in program order.  Laundering the offset through an empty asm makes the compiler
            // treat it as possibly aliasing (it would otherwise prove g + 1 != g for this thread and reorder)
            // without fencing the loads of the other LDS objects.
            int ou = off[k] + 1;
            asm volatile("" : "+v"(ou));
            float* gu = acc + ou;
            float y0 = gu[0], y1 = gu[AP];
            gu[0] = y0 + c01[k];
            gu[AP] = y1 + c11[k];
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
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
            float sv = s[t][r] + u0 + fx * (u1 - u0);
            if (last && step * KT + kh * 32 + crow(r, hi) >= d.N) sv = BEVR_NEG_BIG;
            float ds = BEVR_LN2 * fast_exp2(sv) * dp[t][r];
            s[t][r] = ds;
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
      }
#pragma unroll
      for (int t = 0; t < NQ; ++t) dq[t] = mma_acc_b(ktf, s[t], dq[t]);
    }

    if (step + 1 < n_step) stage_store(buf ^ 1, step + 1);
    __syncthreads();
  }
  if (acc_live) flush(rg);

  // ---- merge the two key halves' dQ partial sums (waves w and w + 4) through LDS, then store ------------
  float* xch = acc_all;   // the accumulation windows are flushed and dead now: 4 waves x 2 x 16 x 64 floats = 32 KiB
  static_assert(NWAVE * (WACC + 3) * AP * 4 >= 4 * NQ * 16 * 64 * 4, "exchange area");
  __syncthreads();   // nobody reads the staging buffers or the table window any more
  if (kh == 1) {
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[((cp * NQ + t) * 16 + r) * 64 + lane] = dq[t][r];
  }
  __syncthreads();
  if (kh == 0) {
    float* dQh = dQ + ((size_t)ph * Mp) * 32;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      if (!live[t]) continue;
      size_t mq = (size_t)jcol[t] * d.Sp + i0 + lq;
      float* row = dQh + mq * 32;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = dq[t][4 * g4 + k] + xch[((cp * NQ + t) * 16 + 4 * g4 + k) * 64 + lane];
        *reinterpret_cast<f32x4*>(row + 8 * g4 + 4 * hi) = v;
      }
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* K, const void* Kt, const void* V, const float* key_a,
           const float* key_b, const float* table_pair, const void* dO, const float* LSE, const float* delta,
           float* dQ, float* dtable, hipStream_t st) {
  const int n_rb = d.Sp / 32, n_cb = (d.S + 4 * NQ - 1) / (4 * NQ);
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_rb * n_cb;
  hipLaunchKernelGGL((attn_bwd_q_kernel<PREC>), dim3(grid), dim3(TQ), 0, st, d, (const char*)Q, (const char*)K,
                     (const char*)Kt, (const char*)V, key_a, key_b, (const char*)table_pair, (const char*)dO, LSE,
                     delta, dQ, dtable);
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
    return launch<BEVR_PREC_BF16>(*d, Q, K, Kt, V, key_a, key_b, table_pair, dO, LSE, delta, dQ, dtable, st);
  return launch<BEVR_PREC_F32>(*d, Q, K, Kt, V, key_a, key_b, table_pair, dO, LSE, delta, dQ, dtable, st);
}
