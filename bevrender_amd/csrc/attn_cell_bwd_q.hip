// Attention backward, query side, over a CELL-SORTED key segment (attn_cell.h): dQ and the rpe-table gradient.
// Counterpart of attn_bwd_q.hip for keys whose 32-key tiles fit one table chunk; same operand layouts, same gradient
// semantics (include/bevrender_hip.h), same work split as attn_cell_fwd.hip (workgroup = one BEV column: one wave per
// 32-row block + a producer wave that stages the next step and builds its weights, their transposes and the tile geometry).
//
// The table gradient of a tile is the transpose of its bias product:
//   dTsh[k'][i] += sum_n W[n][k'] dS^T[n][i]          (k' = chunk cell, i = BEV row of the lane)
// one MFMA pair with dS^T taken straight from the accumulator as the B operand (as dQ is), into 8 live accumulator
// registers per lane that persist for as long as consecutive tiles share the chunk origin -- no per-pair LDS atomics,
// no fixed point.  When the origin changes (3 % of the tiles of a cell-sorted segment) the wave adds its 16 cells x 32
// rows to the table gradient in HBM with float atomics (32 consecutive rows of one table column per instruction and
// lane half: two contiguous 128-byte runs).
// VALU-bound like the forward: zero-accumulator MFMA chains, the row constants (-LSE, -delta) and P (dP - delta) as
// packed f32 arithmetic.
// Tiles that do not fit one chunk are left to a second, SLOW pass of the same kernel (per-pair gathers from the table in
// global memory, float atomics): it lists its column's slow tiles, exits at once when there is none, and stages only
// the listed tiles' steps.
// dQ and dtable are ACCUMULATED: the region kernel (attn_bwd_q.hip) may have written the other key segment's share.
#include <type_traits>
#include "attn_cell.h"

namespace {

template <int PREC> struct LdsCQ {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int R_STRIDE = 32 * EB + 16;     // K, V rows
  static constexpr int T_STRIDE = KT * EB + 16;     // K^T channel rows
  static constexpr int R_BYTES = KT * R_STRIDE;
  static constexpr int T_BYTES = 32 * T_STRIDE;
  static constexpr int KW_BYTES = KT * 16;
  static constexpr int WL = 8 * EB;                 // one lane's chunk operand (lane = key)
  static constexpr int W_BYTES = 2 * 64 * WL;
  static constexpr int WT_STRIDE = 32 * EB + 16;    // W^T rows: 32 key positions (perm32 order) per chunk cell
  static constexpr int WT_TILE = 32 * WT_STRIDE;    // 32 rows: cells 16..31 stay zero (one chunk = 16 cells)
  static constexpr int CT_BYTES = 2 * 16;
  static constexpr int OFF_V = R_BYTES, OFF_KT = 2 * R_BYTES, OFF_KW = 2 * R_BYTES + T_BYTES, OFF_W = OFF_KW + KW_BYTES,
                       OFF_WT = OFF_W + W_BYTES, OFF_CT = OFF_WT + 2 * WT_TILE;
  static constexpr int OFF_DUMMY = OFF_CT + CT_BYTES;   // 16 B that idle staging threads write (keeps staging branch-free)
  static constexpr int BUF = OFF_DUMMY + 16;
  static constexpr int RCH_ROW = 32 * EB / 16;
  static constexpr int TCH_ROW = KT * EB / 16;
  static constexpr int CH = KT * RCH_ROW;
  static constexpr int NCH = 3 * CH + KT;
  static constexpr int NST = is16(PREC) ? 2 : 4;
  static constexpr int QCH = 32 * EB / 16 / 2;      // 16-B chunks of one lane's fragment: 2 (bf16) / 4 (f32)
  static constexpr int QSLOT = 2 * QCH * 1024;      // per wave: Q then dO, each [chunk][lane]
};

template <int PREC>
__device__ __forceinline__ void chunk_map_q(int g, const char* Kh, const char* Vh, const char* Kth, const char* kws,
                                            int Np, const char*& src, int& inc, int& dst) {
  typedef LdsCQ<PREC> L;
  constexpr int EB = L::EB;
  if (g < 2 * L::CH) {
    const int kind = g / L::CH, ci = g % L::CH;
    src = (kind ? Vh : Kh) + (size_t)ci * 16;
    inc = L::CH * 16;
    dst = kind * L::R_BYTES + (ci / L::RCH_ROW) * L::R_STRIDE + (ci % L::RCH_ROW) * 16;
  } else if (g < 3 * L::CH) {
    const int ci = g - 2 * L::CH;
    src = Kth + ((size_t)(ci / L::TCH_ROW) * Np) * EB + (ci % L::TCH_ROW) * 16;
    inc = KT * EB;
    dst = L::OFF_KT + (ci / L::TCH_ROW) * L::T_STRIDE + (ci % L::TCH_ROW) * 16;
  } else {
    const int ci = g - 3 * L::CH;
    src = kws + (size_t)ci * 16;
    inc = KT * 16;
    dst = L::OFF_KW + ci * 16;
  }
}

// acc1 += A1 * X, acc2 += A2 * X with X an accumulator-layout tile used as the B operand of both (bevr_common.h:
// mma_acc_b), converted to bf16 once.
template <int PREC, std::enable_if_t<is16(PREC), int> = 0>
__device__ __forceinline__ void mma_acc_b2(const Frag<PREC>& a1, const Frag<PREC>& a2, const f32x16& x, f32x16& acc1,
                                           f32x16& acc2) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = Half<PREC>::pack2(x[8 * s + 2 * k], x[8 * s + 2 * k + 1]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, w);
    acc1 = Half<PREC>::mfma(a1.v[s], b, acc1);
    acc2 = Half<PREC>::mfma(a2.v[s], b, acc2);
  }
}
template <int PREC, std::enable_if_t<!is16(PREC), int> = 0>
__device__ __forceinline__ void mma_acc_b2(const Frag<PREC>& a1, const Frag<PREC>& a2, const f32x16& x, f32x16& acc1,
                                           f32x16& acc2) {
  acc1 = mma_acc_b(a1, x, acc1);
  acc2 = mma_acc_b(a2, x, acc2);
}

// MAXT: the launch bound.  1024 threads (BEV sides up to 512) cap a wave at 128 registers; the f32-layout modes need
// more than that (fragments twice as wide) and get a 512-thread instantiation (BEV sides up to 256) without spills.
template <int PREC, bool SLOW, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_cell_bwd_q_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ Kt,
    const char* __restrict__ V, const char* __restrict__ key_ws, const char* __restrict__ table_pair,
    const char* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ delta,
    const float* __restrict__ grad_scale, float* __restrict__ dQ, float* __restrict__ dtable) {
  typedef LdsCQ<PREC> L;
  // fp16 mode (include/bevrender_hip.h, grad_scale[2..5]): P' = P 2^kp, dS16 = P' (dP - delta) c2 -- both inside fp16's
  // normal range; dQ and the table gradient are accumulated in dS16 units and unscaled on the way out
  const float kp16 = PREC == BEVR_PREC_F16 ? grad_scale[2] : 0.f, c2_16 = PREC == BEVR_PREC_F16 ? grad_scale[3] : 1.f;
  const float out_scale = BEVR_LN2 * (PREC == BEVR_PREC_F16 ? grad_scale[4] : 1.f);
  constexpr int EB = L::EB;
  // 2 staging buffers | a (Q, dO) fragment slot per wave | (slow pass) the list of this column's slow tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / d.S) * 8 + xcd;
  if (ph >= n_ph) return;
  const int j = slot % d.S;
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int grp = hd / (d.heads / d.groups);
  const int qb = prob / d.q_div;

  const int tid = threadIdx.x, nt = blockDim.x, n_wave = nt >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const int Mp = d.S * d.Sp;
  // fast pass: the LAST wave is the workgroup's PRODUCER (attn_cell_fwd.hip): it stages the next step's keys and builds
  // their weight tiles while the other waves (one per 32-row block) run the tiles; it owns no BEV rows (its row indices
  // alias row block 0 so that the prologue stays in range; it leaves before anything is written)
  const bool producer = !SLOW && wave == n_wave - 1;
  const int i0 = (producer ? 0 : wave) * 32;

  const char* Qh = Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB;
  const char* dOh = dO + ((size_t)ph * Mp) * 32 * EB;
  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = V + ((size_t)ph * d.Np) * 32 * EB;
  const char* Kth = Kt + ((size_t)ph * 32) * d.Np * EB;
  const int pg = prob * d.groups + grp;
  const char* kws = key_ws + (size_t)pg * d.Np * sizeof(KeyW);
  const StepBox* kbox = reinterpret_cast<const StepBox*>(key_ws + key_ws_box_offset(d)) + (size_t)pg * (d.Np / 32);
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const int Hq = d.Hp + 1;
  float* dtb = dtable + (size_t)hd * d.Wp * Hq;
  const int Hp8 = d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const float jrx = (float)j * rx;
  const int n_step = d.Np / KT;

  char* qslot = smem + 2 * L::BUF + wave * L::QSLOT + lane * 16;
  int* slow_list = reinterpret_cast<int*>(smem + 2 * L::BUF + n_wave * L::QSLOT);
  __shared__ int slow_count;
  if constexpr (SLOW) {
    // this column's slow tiles, listed in key order: a flag byte per tile, then wave 0 compacts the flags with ballots
    unsigned char* flag = reinterpret_cast<unsigned char*>(smem);   // the staging buffers are not in use yet
    for (int u = tid; u < 2 * n_step; u += nt) {
      const CellTile c = make_celltile(kbox[u], jrx);
      flag[u] = (c.live && !c.fast) ? 1 : 0;
    }
    __syncthreads();
    if (wave == 0) {
      int cnt = 0;
      for (int b0 = 0; b0 < 2 * n_step; b0 += 64) {
        const bool f = b0 + lane < 2 * n_step && flag[b0 + lane];
        const unsigned long long mask = __ballot(f);
        if (f) slow_list[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = b0 + lane;
        cnt += __popcll(mask);
      }
      if (lane == 0) slow_count = cnt;
    }
    __syncthreads();
    if (slow_count == 0) return;
  }

  // this lane's query; rows past the grid compute on a clamped copy with dO = delta = 0 (their dS is exactly 0)
  const int qrow = i0 + lq;
  const bool live = qrow < d.S;
  const size_t mq = (size_t)j * d.Sp + min(qrow, d.S - 1);
  const float lse = LSE[(size_t)ph * Mp + mq];
  float dlt = delta[(size_t)ph * Mp + mq];
  if (!live) dlt = 0.f;
  // the Q and dO fragments live in LDS (own lanes' data, written and read by this wave only: no barrier), re-read per
  // tile; slot layout [chunk][lane]: consecutive lanes read consecutive 16 B (a per-lane slot of 64 / 128 B put 4 / 8
  // lanes of a ds_read_b128 group on the same banks: measured 65 % of this kernel's LDS cycles)
  auto put_frag = [&](char* dst, const Frag<PREC>& f, bool zero) {
    if constexpr (is16(PREC)) {
      const u32x4 z = {0, 0, 0, 0};
      *reinterpret_cast<u32x4*>(dst) = zero ? z : __builtin_bit_cast(u32x4, f.v[0]);
      *reinterpret_cast<u32x4*>(dst + 1024) = zero ? z : __builtin_bit_cast(u32x4, f.v[1]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(dst + 1024 * k) =
            zero ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{f.v[4 * k], f.v[4 * k + 1], f.v[4 * k + 2], f.v[4 * k + 3]};
    }
  };
  auto get_frag = [&](const char* src, Frag<PREC>& f) {
    if constexpr (is16(PREC)) {
      f.v[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src));
      f.v[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + 1024));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 1024 * k);
        f.v[4 * k] = t[0]; f.v[4 * k + 1] = t[1]; f.v[4 * k + 2] = t[2]; f.v[4 * k + 3] = t[3];
      }
    }
  };
  {
    Frag<PREC> f;
    f.load(Qh + mq * 32 * EB, hi);
    put_frag(qslot, f, false);
    f.load(dOh + mq * 32 * EB, hi);
    put_frag(qslot + L::QCH * 1024, f, !live);
  }

  f32x16 dq, y;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq[r] = 0.f; y[r] = 0.f; }

  CellFrag<PREC> tf;
  int tag_x = 1 << 30, tag_a = 1 << 30;
  if constexpr (is16(PREC)) tf.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  else {
#pragma unroll
    for (int k = 0; k < 8; ++k) tf.v[k] = 0.f;
  }
  // add this wave's chunk of table gradient to HBM: register r of lane (i, hi) is chunk cell crow(r, hi) = 4 c + row,
  // i.e. table entry (tag_x + c, tag_a + row + i0 + i); only cells < 16 exist
  auto flush_y = [&]() {
    if (tag_x != (1 << 30)) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = 2 * (r >> 2) + hi, row = r & 3;
        const int xc = tag_x + c + d.x_off, yr = tag_a + row + i0 + lq + d.y_off;
        const float v = y[r] * out_scale;
        if (v != 0.f && xc >= 0 && xc < d.Wp && yr >= 0 && yr < Hq) atomicAdd(dtb + (size_t)xc * Hq + yr, v);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = 0.f;
  };

  // ---- one tile ------------------------------------------------------------------------------------------------
  // MASKED: the tile may hold padded keys (last step only); its own instantiation (attn_cell_fwd.hip)
  auto tile = [&](auto masked_tag, const char* base, int step, int t, int x0, int a0) {
    constexpr bool last = decltype(masked_tag)::value;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }   // literal-zero accumulators: no register splats
    {
      Frag<PREC> kf, qf;
      kf.load(base + (t * 32 + lq) * L::R_STRIDE, hi);
      get_frag(qslot, qf);
      s = mma_frag(kf, qf, s);   // S^T
    }
    {
      Frag<PREC> vkf, dof;
      vkf.load(base + L::OFF_V + (t * 32 + lq) * L::R_STRIDE, hi);
      get_frag(qslot + L::QCH * 1024, dof);
      dp = mma_frag(vkf, dof, dp);   // dP^T
    }
    if constexpr (!SLOW) {
      if (x0 != tag_x || a0 != tag_a) {   // uniform: new chunk origin
        flush_y();
        tf = cell_table<PREC>(tbl, d, x0, a0 + i0 + lq, hi);
        tag_x = x0;
        tag_a = a0;
      }
      CellFrag<PREC> wf;
      const char* wsrc = base + L::OFF_W + (t * 64 + lane) * L::WL;
      if constexpr (is16(PREC)) {
        wf.v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wsrc));
      } else {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wsrc), w1 = *reinterpret_cast<const f32x4*>(wsrc + 16);
        wf.v[0] = w0[0]; wf.v[1] = w0[1]; wf.v[2] = w0[2]; wf.v[3] = w0[3];
        wf.v[4] = w1[0]; wf.v[5] = w1[1]; wf.v[6] = w1[2]; wf.v[7] = w1[3];
      }
      s = mma_cell(wf, tf, s);
      if constexpr (last) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (step * KT + t * 32 + crow(r, hi) >= d.N) ? BEVR_NEG_BIG : s[r];
      }
      // P = exp2(S - LSE), dS = P (dP - delta), two rows per instruction; ln2 applied on the way out
      const f32x2 nl = {kp16 - lse, kp16 - lse}, nd = {-dlt, -dlt};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sh = f32x2{s[r], s[r + 1]} + nl;
        const f32x2 pp = {fast_exp2(sh[0]), fast_exp2(sh[1])};
        f32x2 ds = pp * (f32x2{dp[r], dp[r + 1]} + nd);
        if constexpr (PREC == BEVR_PREC_F16) ds *= f32x2{c2_16, c2_16};
        s[r] = ds[0];
        s[r + 1] = ds[1];
      }
      Frag<PREC> ktf, wtf;
      load_perm(ktf, base + L::OFF_KT + lq * L::T_STRIDE + t * 32 * EB, hi);
      load_perm(wtf, base + L::OFF_WT + t * L::WT_TILE + lq * L::WT_STRIDE, hi);
      mma_acc_b2(ktf, wtf, s, dq, y);
    } else {
      // per-pair path: gathers from the table in global memory, float atomics into the table gradient
      const KeyW* kwl = reinterpret_cast<const KeyW*>(base + L::OFF_KW);
      const int rowoff = (i0 + lq) * 8;
      const int xoffHp = d.x_off * d.Hp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const KeyW c = kwl[t * 32 + crow(r, hi)];
        const float wy0 = 1.0f - c.fy;
        const float tx = jrx + c.b;
        const float xf = floorf(tx);
        const float fx = tx - xf;
        const int xi = (int)xf;
        const unsigned off = (unsigned)(xi * Hp8 + c.aoff + rowoff);
        const f32x2 t0 = *reinterpret_cast<const f32x2*>(tbl + off);
        const f32x2 t1 = *reinterpret_cast<const f32x2*>(tbl + off + Hp8);
        const float u0 = t0[0] * wy0 + t0[1] * c.fy;
        const float u1 = t1[0] * wy0 + t1[1] * c.fy;
        float sv = s[r] + (kp16 - lse) + u0 + fx * (u1 - u0);
        if (last && step * KT + t * 32 + crow(r, hi) >= d.N) sv = BEVR_NEG_BIG;
        const float ds = fast_exp2(sv) * (dp[r] - dlt) * c2_16;
        s[r] = ds;
        if (ds != 0.f) {
          const int yi = (c.aoff >> 3) - xoffHp + i0 + lq;
          float* g0 = dtb + (size_t)(xi + d.x_off) * Hq + yi;
          const float w0 = out_scale * ds * (1.0f - fx), w1 = out_scale * ds * fx;
          atomicAdd(g0, w0 * wy0);
          atomicAdd(g0 + 1, w0 * c.fy);
          atomicAdd(g0 + Hq, w1 * wy0);
          atomicAdd(g0 + Hq + 1, w1 * c.fy);
        }
        if ((r & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight (register budget)
      }
      Frag<PREC> ktf;
      load_perm(ktf, base + L::OFF_KT + lq * L::T_STRIDE + t * 32 * EB, hi);
      dq = mma_acc_b(ktf, s, dq);
    }
  };

  // ---- staging helpers -----------------------------------------------------------------------------------------
  auto stage_direct = [&](char* base, int step, int g0) {
    for (int g = g0; g < L::NCH; g += nt) {
      const char* src;
      int inc, dst;
      chunk_map_q<PREC>(g, Kh, Vh, Kth, kws, d.Np, src, inc, dst);
      *reinterpret_cast<u32x4*>(base + dst) = *reinterpret_cast<const u32x4*>(src + (size_t)step * inc);
    }
  };

  if constexpr (SLOW) {
    const int n_slow = slow_count;
    for (int u = 0; u < n_slow; ++u) {
      const int tile_id = slow_list[u];
      const int step = tile_id >> 1, t = tile_id & 1;
      __syncthreads();                  // every wave is done with the previous tile's buffer
      stage_direct(smem, step, tid);
      __syncthreads();
      if (step == n_step - 1 && d.N < d.Np) tile(std::true_type{}, smem, step, t, 0, 0);
      else tile(std::false_type{}, smem, step, t, 0, 0);
    }
  } else {
    // weights of tile t of a step, by this wave (lane & 31 = key): W for the bias product (lane = key), its transpose for
    // the table gradient (rows = chunk cells, 32 key positions in perm32 order), and the tile's geometry
    auto build_w = [&](int buf, int step, int t, const KeyW& kw, const StepBox& sb) {
      const CellTile ct = make_celltile(sb, jrx);
      float tcol, trow;
      cell_coords(kw, jrx, ct.x0, step * KT + t * 32 + lq >= d.N, tcol, trow);
      const CellFrag<PREC> w = cell_weights<PREC>(tcol, trow, hi);
      char* bb = smem + buf * L::BUF;
      char* dst = bb + L::OFF_W + (t * 64 + lane) * L::WL;
      char* wt = bb + L::OFF_WT + t * L::WT_TILE + perm32(lq) * EB;
      if constexpr (is16(PREC)) {
        const u32x4 wv = __builtin_bit_cast(u32x4, w.v);
        *reinterpret_cast<u32x4*>(dst) = wv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // cell k' = 8 hi + e
          const unsigned short hv = (unsigned short)(e & 1 ? wv[e >> 1] >> 16 : wv[e >> 1] & 0xffffu);
          *reinterpret_cast<unsigned short*>(wt + (8 * hi + e) * L::WT_STRIDE) = hv;
        }
      } else if constexpr (PREC == BEVR_PREC_BF16X3) {
        // w holds the two bf16 planes (attn_cell.h); its transpose goes into the split perm32 block format
        // (bevr_common.h): f32 position q = 16 s + 8 h + j -> hi plane chunk 2 h + s, lo plane chunk 4 + 2 h + s
        *reinterpret_cast<f32x4*>(dst) = f32x4{w.v[0], w.v[1], w.v[2], w.v[3]};
        *reinterpret_cast<f32x4*>(dst + 16) = f32x4{w.v[4], w.v[5], w.v[6], w.v[7]};
        const int q = perm32(lq);
        char* wq = bb + L::OFF_WT + t * L::WT_TILE + (2 * ((q >> 3) & 1) + (q >> 4)) * 16 + 2 * (q & 7);
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {   // cell k' = 2 t8 + hi
          const unsigned hw = __builtin_bit_cast(unsigned, w.v[t8 >> 1]), lw = __builtin_bit_cast(unsigned, w.v[4 + (t8 >> 1)]);
          char* row = wq + (2 * t8 + hi) * L::WT_STRIDE;
          *reinterpret_cast<unsigned short*>(row) = (unsigned short)(t8 & 1 ? hw >> 16 : hw & 0xffffu);
          *reinterpret_cast<unsigned short*>(row + 64) = (unsigned short)(t8 & 1 ? lw >> 16 : lw & 0xffffu);
        }
      } else {
        *reinterpret_cast<f32x4*>(dst) = f32x4{w.v[0], w.v[1], w.v[2], w.v[3]};
        *reinterpret_cast<f32x4*>(dst + 16) = f32x4{w.v[4], w.v[5], w.v[6], w.v[7]};
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)   // cell k' = 2 t8 + hi
          *reinterpret_cast<float*>(wt + (2 * t8 + hi) * L::WT_STRIDE) = w.v[t8];
      }
      if (lane == 0) *reinterpret_cast<CellTile*>(bb + L::OFF_CT + t * 16) = ct;
    };
    auto load_kw = [&](int step, int t) {
      return *reinterpret_cast<const KeyW*>(kws + ((size_t)step * KT + t * 32 + lq) * sizeof(KeyW));
    };
    // W^T rows 16..31 of every tile stay zero (a chunk has 16 cells; the MFMA tile has 32 rows)
    for (int u = tid; u < 2 * 2 * 16 * (L::WT_STRIDE / 16); u += nt) {
      const int q16 = u % (L::WT_STRIDE / 16), row = (u / (L::WT_STRIDE / 16)) % 16, tb = u / (16 * (L::WT_STRIDE / 16));
      *reinterpret_cast<u32x4*>(smem + (tb >> 1) * L::BUF + L::OFF_WT + (tb & 1) * L::WT_TILE +
                                (16 + row) * L::WT_STRIDE + q16 * 16) = u32x4{0, 0, 0, 0};
    }
    // a last step with padded keys is peeled off behind the loop (its masked tile body inside the loop cost every tile
    // registers or hoisted compares)
    const int n_main = d.N < d.Np ? n_step - 1 : n_step;
    if (producer) {
      // ---- the producer wave: global -> registers -> LDS one step ahead, and the weight tiles / geometry of the step
      constexpr int NSTP = (L::NCH + 63) / 64;        // 16-byte chunks per lane and step
      u32x4 st[NSTP];
      const char* st_src[NSTP];
      int st_inc[NSTP], st_dst[NSTP];
#pragma unroll
      for (int k = 0; k < NSTP; ++k) {
        const int g = lane + 64 * k;
        // lanes beyond the chunk count re-read chunk 0 into a dummy slot: loads and stores stay unconditional
        chunk_map_q<PREC>(g < L::NCH ? g : 0, Kh, Vh, Kth, kws, d.Np, st_src[k], st_inc[k], st_dst[k]);
        if (g >= L::NCH) st_dst[k] = L::OFF_DUMMY;
      }
#pragma unroll
      for (int k = 0; k < NSTP; ++k) *reinterpret_cast<u32x4*>(smem + st_dst[k]) = gload16(st_src[k]);
      build_w(0, 0, 0, load_kw(0, 0), kbox[0]);
      build_w(0, 0, 1, load_kw(0, 1), kbox[1]);
      __syncthreads();
      for (int step = 0; step < n_main; ++step) {
        if (step + 1 < n_step) {
#pragma unroll
          for (int k = 0; k < NSTP; ++k) {
            st_src[k] += st_inc[k];
            st[k] = gload16(st_src[k]);
          }
          const KeyW kw0 = load_kw(step + 1, 0), kw1 = load_kw(step + 1, 1);
          const int nbuf = (step + 1) & 1;
          build_w(nbuf, step + 1, 0, kw0, kbox[2 * (step + 1)]);
          build_w(nbuf, step + 1, 1, kw1, kbox[2 * (step + 1) + 1]);
          char* nb = smem + nbuf * L::BUF;
#pragma unroll
          for (int k = 0; k < NSTP; ++k) *reinterpret_cast<u32x4*>(nb + st_dst[k]) = st[k];
        }
        __syncthreads();
      }
      return;
    }
    // ---- the row-block waves ----
    __syncthreads();   // step 0 is staged
    for (int step = 0; step < n_main; ++step) {
      const char* base = smem + (step & 1) * L::BUF;
#pragma unroll 1
      for (int t = 0; t < 2; ++t) {
        const u32x4 cw = *reinterpret_cast<const u32x4*>(base + L::OFF_CT + t * 16);   // geometry, by the producer
        const int live = __builtin_amdgcn_readfirstlane((int)cw[0]), fast = __builtin_amdgcn_readfirstlane((int)cw[1]);
        if (!live || !fast) continue;   // nothing to do / the slow pass's tile (uniform)
        tile(std::false_type{}, base, step, t, __builtin_amdgcn_readfirstlane((int)cw[2]),
             __builtin_amdgcn_readfirstlane((int)cw[3]));
      }
      __syncthreads();
    }
    if (n_main < n_step) {   // the peeled last step: padded keys masked
      const char* base = smem + (n_main & 1) * L::BUF;
#pragma unroll 1
      for (int t = 0; t < 2; ++t) {
        const u32x4 cw = *reinterpret_cast<const u32x4*>(base + L::OFF_CT + t * 16);
        const int live = __builtin_amdgcn_readfirstlane((int)cw[0]), fast = __builtin_amdgcn_readfirstlane((int)cw[1]);
        if (!live || !fast) continue;
        tile(std::true_type{}, base, n_main, t, __builtin_amdgcn_readfirstlane((int)cw[2]),
             __builtin_amdgcn_readfirstlane((int)cw[3]));
      }
    }
    flush_y();
  }

  // ---- dQ: added to what is there (the other key segment's share, or the caller's zeros) -----------------------
  if (live) {
    float* row = dQ + ((size_t)ph * Mp + (size_t)j * d.Sp + qrow) * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(row + 8 * g4 + 4 * hi);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += out_scale * dq[4 * g4 + k];
      *reinterpret_cast<f32x4*>(row + 8 * g4 + 4 * hi) = v;
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* K, const void* Kt, const void* V, const void* key_ws,
           const float* table_pair, const void* dO, const float* LSE, const float* delta, const float* gs, float* dQ,
           float* dtable, hipStream_t st) {
  typedef LdsCQ<PREC> L;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * d.S;
  const int n_rb = d.Sp / 32;                 // one wave per 32-row block of the column ...
  const int n_fast = n_rb + 1;                // ... + the producer wave (fast pass)
  if (n_fast > 16) return BEVR_E_SHAPE;
  const size_t lds = 2 * L::BUF + (size_t)n_fast * L::QSLOT;
  const size_t lds_slow = 2 * L::BUF + (size_t)n_rb * L::QSLOT + (size_t)(d.Np / 32) * 4;
  if (lds > 160 * 1024 || lds_slow > 160 * 1024) return BEVR_E_SHAPE;
  if (!is16(PREC) && 64 * n_fast <= 512)
    hipLaunchKernelGGL((attn_cell_bwd_q_kernel<PREC, false, (is16(PREC) ? 1024 : 512)>), dim3(grid), dim3(64 * n_fast), lds, st,
                       d, (const char*)Q, (const char*)K, (const char*)Kt, (const char*)V, (const char*)key_ws,
                       (const char*)table_pair, (const char*)dO, LSE, delta, gs, dQ, dtable);
  else
    hipLaunchKernelGGL((attn_cell_bwd_q_kernel<PREC, false, 1024>), dim3(grid), dim3(64 * n_fast), lds, st, d, (const char*)Q,
                       (const char*)K, (const char*)Kt, (const char*)V, (const char*)key_ws, (const char*)table_pair,
                       (const char*)dO, LSE, delta, gs, dQ, dtable);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if (!is16(PREC) && 64 * n_rb <= 512)
    hipLaunchKernelGGL((attn_cell_bwd_q_kernel<PREC, true, (is16(PREC) ? 1024 : 512)>), dim3(grid), dim3(64 * n_rb), lds_slow, st,
                       d, (const char*)Q, (const char*)K, (const char*)Kt, (const char*)V, (const char*)key_ws,
                       (const char*)table_pair, (const char*)dO, LSE, delta, gs, dQ, dtable);
  else
    hipLaunchKernelGGL((attn_cell_bwd_q_kernel<PREC, true, 1024>), dim3(grid), dim3(64 * n_rb), lds_slow, st, d,
                       (const char*)Q, (const char*)K, (const char*)Kt, (const char*)V, (const char*)key_ws,
                       (const char*)table_pair, (const char*)dO, LSE, delta, gs, dQ, dtable);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_cell_bwd_q(const bevr_attn_desc* d, const void* Q, const void* K, const void* Kt, const void* V,
                                    const void* key_ws, const float* table_pair, const void* dO, const float* LSE,
                                    const float* delta, const float* grad_scale, float* dQ, float* dtable,
                                    void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !K || !Kt || !V || !key_ws || !table_pair || !dO || !LSE || !delta || !dQ || !dtable ||
      (d->precision == BEVR_PREC_F16 && !grad_scale))
    return BEVR_E_NULL;
  if (d->Sp > 512) return BEVR_E_SHAPE;
  if (!bevr_aligned16(Q) || !bevr_aligned16(K) || !bevr_aligned16(Kt) || !bevr_aligned16(V) || !bevr_aligned16(dO) ||
      !bevr_aligned16(dQ) || !bevr_aligned16(table_pair) || !bevr_aligned16(key_ws))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st);
  if (d->precision == BEVR_PREC_F16)
    return launch<BEVR_PREC_F16>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st);
  if (d->precision == BEVR_PREC_BF16X3)
    return launch<BEVR_PREC_BF16X3>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st);
  return launch<BEVR_PREC_F32>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st);
}
