// Attention backward, query side: dQ and the rpe-table gradient.
// Same orientation, LDS table region and XCD mapping as the forward (attn_fwd.hip): S^T[key][query] tiles with
// the query on the lane, so
//   * P^T = exp2(S^T - LSE[q]) and dS^T = ln2 * P^T * (dP^T - delta[q]) need only lane-local constants
//     (preloaded into the accumulators of the two MFMA chains),
//   * dQ^T[c][q] += K^T[c][key] dS^T[key][q] takes dS^T straight from the accumulator (B operand),
//   * the table gradient of a (query tile x key step) block lands inside the same box the bias was read
//     from: it is accumulated in ONE workgroup-shared LDS window of the region's shape with fixed-point
//     integer atomics (ds_add_u32 in bf16 mode, ds_add_u64 in f32 mode) and flushed to HBM with contiguous
//     float atomics only when
//     the region moves.  Measured on gfx950: LDS float atomics cost ~160 cycles per wave instruction and a
//     plain read-modify-write needs per-wave windows, lane regrouping and ordering that pushed the kernel
//     into scratch spills; the integer adds are cheap, order-free and make the window sums bit-reproducible.
//     The fixed-point unit is 2^-30 (f32 mode) / 2^-26 (bf16 mode) of a bound on |dS| handed in by the caller
//     (grad_scale); see AccCell.
//     Steps whose box does not fit scatter straight to global memory with float atomics.
// Workgroup = 8 waves on one 32-row x 8-column query tile (wave = column; both 32-key halves of a 64-key step in
// turn), TWO workgroups per CU (4 waves per SIMD, <= 128 registers): the kernel's phases -- staging, MFMA, the
// LDS-issue-bound bias/gradient loop, region moves -- are separated by barriers inside a workgroup, so a second
// resident workgroup is what keeps the LDS pipe busy during the other's non-LDS phases.  One staging buffer
// (registers decouple the global loads) keeps a workgroup under 80 KB of LDS.
// Recomputes S from Q, K and the bias instead of storing any (M x N) tensor.
// Gradient semantics: see include/bevrender_hip.h (log2-domain inputs as handed in).
#include <type_traits>
#include "attn_tile.h"

#ifdef BEVR_PROF
__device__ unsigned long long bevr_prof[16];
extern "C" int bevr_debug_prof(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(bevr_prof), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bevr_prof), 16 * 8);
}
__device__ __forceinline__ unsigned long long prof_now(float dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}
#define PROF_T(var) const unsigned long long var = prof_now(0.f)
#define PROF_TD(var, dep) const unsigned long long var = prof_now(dep)
#define PROF_ADD(i, v) pacc[i] += (v)
#else
#define PROF_T(var)
#define PROF_TD(var, dep)
#define PROF_ADD(i, v)
#endif

namespace {

constexpr int TQ = 512;   // threads per workgroup
constexpr int NWAVE = TQ / 64;
constexpr int NCOL = 8;   // query columns per workgroup (one per wave)

template <int PREC> struct LdsQ {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int CAP = region_cap_bwd_q(PREC);   // region capacity (table columns)
  static constexpr int R_STRIDE = 32 * EB + 16;   // row-layout tiles (K, V): bytes per key row
  static constexpr int T_STRIDE = KT * EB + 16;   // transposed tile (Kt): bytes per channel row
  static constexpr int R_BYTES = KT * R_STRIDE;
  static constexpr int T_BYTES = 32 * T_STRIDE;
  static constexpr int C_BYTES = KT * 16 + 32;
  static constexpr int BUF = 2 * R_BYTES + T_BYTES + C_BYTES;
  static constexpr int ENT = is16(PREC) ? 4 : 8;   // table-window entry: (T[y], T[y+1]) as a 16-bit pair / f32x2
  static constexpr int WCOLS = CAP + 2;           // + two "kill" columns of -1e30: where masked keys point their taps
  static constexpr int WIN = WCOLS * WIN_PITCH * ENT;
  static constexpr int CELLS = CAP * WIN_PITCH;   // accumulation window: one 64-bit cell per table entry (+ kill columns)
  static constexpr int PCK = NWAVE * 32 * (is16(PREC) ? 16 : 32);   // per wave: one 32-key half at a time
  // bf16 mode: the tile's dO fragments live in LDS (fragment order, re-read every step) instead of 8 registers
  static constexpr int QDO = is16(PREC) ? NCOL * 2 * 64 * 16 : 16;
  static constexpr int ACCB = 8;   // bytes per accumulation cell (see AccCell)
  static constexpr int TOTAL = BUF + WIN + WCOLS * WIN_PITCH * ACCB + PCK + QDO;
};

constexpr int QROWS = 31;   // query rows per tile: lane 31 of each 32-lane half carries no query (see the kernel header)

// round-to-nearest-even float -> int (v_rndne_f32 + v_cvt_i32_f32), as plain C so that the compiler sees the read.
// Round 1 used inline asm (v_cvt_rpi_i32_f32, one instruction).  Inline asm is opaque to the hazard recognizer: fed
// directly by a v_dot2c_f32_bf16 result (two instructions earlier in the stream) it read a stale register and the
// bf16-mode table gradient came out 65 % wrong while the same arithmetic through v_fma_f32 was right
// (tools/micro/dot2_test.hip shows the instruction itself is exact).
__device__ __forceinline__ int cvt_rpi(float x) { return (int)__builtin_rintf(x); }

// the value of the lane below (lane - 1) across the whole wave; lane 0 receives 0 (v_mov_b32_dpp wave_shr:1)
__device__ __forceinline__ float lane_below(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}

// Fixed-point accumulation cell: 64-bit in both precision modes.
//   unit = ln2 * bound * 2^-30, bound >= max |P (dP - delta)| over all pairs (handed in by the caller as grad_scale);
//   one contribution converts to a 32-bit integer (v_cvt_rpi, round to nearest: truncation would bias the sum of
//   many small same-sign contributions) and is sign-extended into the 64-bit cell, so a cell can take 2^32
//   contributions of the largest possible size before it wraps -- more than a launch has pairs per cell.
//   32-bit cells (round 1, bf16 mode: ds_add_u32 is 4.4 clk, ds_add_u64 6.3) cannot hold both ends: a cell of the
//   pinned-key box receives ~10^5 contributions per region, so a unit that is safe against wrap-around is
//   ~bound * 2^-14, far above a typical contribution (P ~ 1/N): at S = 200 the table gradient came out 67 % wrong
//   against the f32 mode (tests/test_gpu_fullsize.py).  Native LDS float atomics are no way out on gfx950 either:
//   ds_add_f32 / ds_pk_add_bf16 retire ~3 clk per active LANE (193 clk per wave instruction; tools/micro/lds_bench.hip).
struct AccCell {
  typedef unsigned long long type;
  static __device__ __forceinline__ type from_int(int v) {
    return ((unsigned long long)(unsigned)(v >> 31) << 32) | (unsigned)v;
  }
  static __device__ __forceinline__ type from(float x) { return from_int(cvt_rpi(x)); }
  // whole 64-bit value at once: converting the halves separately rounds the low word of a small NEGATIVE sum
  // (hi = -1, lo = 2^32 - k) to a multiple of 256 units before the halves cancel -- up to 128 units of error per
  // flushed cell, which over the thousands of flushes a table entry receives was 1.3 % of the S = 200 table gradient
  static __device__ __forceinline__ float to_float(type v) {
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    if (hi == (lo >> 31)) return (float)lo;   // fits 32 bits (nearly always): one conversion instead of the emulated 64-bit one
    return (float)(long long)v;
  }
};

template <int PREC>
__global__ __launch_bounds__(TQ, is16(PREC) ? 4 : 2) void attn_bwd_q_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ Kt,
    const char* __restrict__ V, const char* __restrict__ key_ws,
    const char* __restrict__ table_pair, const char* __restrict__ dO, const float* __restrict__ LSE,
    const float* __restrict__ delta, const float* __restrict__ grad_scale, float* __restrict__ dQ,
    float* __restrict__ dtable BEVR_DROP_PARAMS) {
  typedef LdsQ<PREC> L;
  constexpr int EB = L::EB;
  constexpr int CAP = L::CAP;
  constexpr int ENT = L::ENT;
  static_assert(L::TOTAL <= (is16(PREC) ? 80 : 160) * 1024, "LDS budget");
  static_assert(NCOL == QCOLS, "group_width() assumes this tile width");
  // separate LDS objects: loads of the staged tiles / table window may be scheduled across the window atomics
  __shared__ __attribute__((aligned(16))) char smem[L::BUF];
  // 256-byte aligned: ds_read2st64_b32 counts its two offsets in units of 256 bytes, so a remainder of the window's
  // base address would cost one v_add per key row
  __shared__ __attribute__((aligned(256))) char win[L::WIN];
  typedef AccCell Acc;
  typedef typename Acc::type acc_t;
  __shared__ __attribute__((aligned(16))) acc_t accw[L::WCOLS * WIN_PITCH];
  typedef ColKeyT<PREC> CK;
  __shared__ __attribute__((aligned(16))) CK pck_all[NWAVE * 32];
  __shared__ __attribute__((aligned(16))) char qdo[L::QDO];

  const int n_rb = (d.S + QROWS - 1) / QROWS;
  const int n_cb = (d.S + NCOL - 1) / NCOL;
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
  const int col = wave;
  CK* pck = pck_all + wave * 32;
  const int Mp = d.S * d.Sp;
  const int i0 = rb * QROWS;

  const char* Qh = Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB;
  const char* dOh = dO + ((size_t)ph * Mp) * 32 * EB;
  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = V + ((size_t)ph * d.Np) * 32 * EB;
  const char* Kth = Kt + ((size_t)ph * 32) * d.Np * EB;
  const int pg = prob * d.groups + grp;
  const KeyW* kws = reinterpret_cast<const KeyW*>(key_ws) + (size_t)pg * d.Np;
  const StepBox* kbox = reinterpret_cast<const StepBox*>(key_ws + key_ws_box_offset(d)) + (size_t)pg * (d.Np / 32);
  const StepBox* gbox = reinterpret_cast<const StepBox*>(key_ws + key_ws_gbox_offset(d)) + (size_t)pg * (d.Np / 32) * N_GROUP;
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  float* dtb = dtable + (size_t)hd * d.Wp * (d.Hp + 1);
  const int Hp8 = d.Hp * 8;
  const int Hq = d.Hp + 1;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int j_first = cb * NCOL;
  const int j_last = min(j_first + NCOL - 1, d.S - 1);
  const float jrx_lo = (float)j_first * rx, jrx_hi = (float)j_last * rx;
  // dS = ln2 * P * (dP - delta).  The fixed-point scale 2^e of the table-gradient cells (a power of two: exact) is
  // folded into dO and delta when they are loaded, so P (dP - delta) comes out of the loop already in cell units;
  // the ln2 and 2^-e are applied once per flushed cell and once per dQ element.
  const float gscale = grad_scale[0], ginv = grad_scale[1] * BEVR_LN2;
  // fp16 mode (include/bevrender_hip.h, grad_scale[2..5]): dO and delta are NOT scaled (2^30 / bound does not fit fp16);
  // P' = P 2^kp and dS16 = P' (dP - delta) c2 stay inside fp16's normal range, the fixed-point cells count
  // dS16 * cfix = dS * s, and dQ leaves in dS16 units
  const float kp16 = PREC == BEVR_PREC_F16 ? grad_scale[2] : 0.f, c2_16 = PREC == BEVR_PREC_F16 ? grad_scale[3] : 1.f;
  const float cfix = PREC == BEVR_PREC_F16 ? grad_scale[0] * grad_scale[4] : 1.f;
  const float dq_scale = PREC == BEVR_PREC_F16 ? grad_scale[4] * BEVR_LN2 : ginv;

  // this wave's query column; this lane's query row.  Lanes 0..30 of each half carry the tile's 31 queries, lane 31
  // none: its slot is the 32nd table row the tile's taps reach (query 30's lower tap), which lets every lane add the
  // PRE-SUMMED contribution of its own upper tap and the lower tap of the lane below -- one atomic per table column
  // and key instead of two.  A lane without a query (lane 31, rows past the grid, columns past the grid) computes on
  // a clamped copy of a real query with dO = delta = 0: its dS is exactly 0.
  const int jcol = j_first + col;
  const int qrow = i0 + lq;
  const bool live = jcol < d.S && lq < QROWS && qrow < d.S;
  const int jc = min(jcol, d.S - 1);
  const float jrx = (float)jc * rx;
  const size_t mq = (size_t)jc * d.Sp + min(qrow, d.S - 1);
  Frag<PREC> qf, dof;
  qf.load(Qh + mq * 32 * EB, hi);
  dof.load(dOh + mq * 32 * EB, hi);
  const float lse = LSE[(size_t)ph * Mp + mq];
  float dlt = delta[(size_t)ph * Mp + mq];
  if (!live) {
    if constexpr (is16(PREC)) {
      dof.v[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      dof.v[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) dof.v[k] = 0.f;
    }
    dlt = 0.f;
  }
  if constexpr (PREC != BEVR_PREC_F16) dlt *= gscale;
  if constexpr (PREC == BEVR_PREC_BF16) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 w = __builtin_bit_cast(u32x4, dof.v[h]);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        w[k] = pack_bf16x2(__builtin_bit_cast(float, w[k] << 16) * gscale,
                           __builtin_bit_cast(float, w[k] & 0xffff0000u) * gscale);
      dof.v[h] = __builtin_bit_cast(bf16x8, w);
    }
  } else if constexpr (PREC == BEVR_PREC_BF16X3) {   // both planes; gscale is a power of two: exact
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const unsigned w = __builtin_bit_cast(unsigned, dof.v[k]);
      dof.v[k] = __builtin_bit_cast(float, pack_bf16x2(__builtin_bit_cast(float, w << 16) * gscale,
                                                        __builtin_bit_cast(float, w & 0xffff0000u) * gscale));
    }
  } else if constexpr (!is16(PREC)) {
#pragma unroll
    for (int k = 0; k < 16; ++k) dof.v[k] *= gscale;
  }
  if constexpr (is16(PREC)) {   // own lanes' data, written and read by this wave only
    u32x4* qd = reinterpret_cast<u32x4*>(qdo) + col * 2 * 64 + lane;
    qd[0] = __builtin_bit_cast(u32x4, dof.v[0]);
    qd[64] = __builtin_bit_cast(u32x4, dof.v[1]);
  }
  const int ilane = i0 + lq;
  const int rowoff = ilane * 8;
  const int xoffHp = d.x_off * d.Hp;

  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;

  constexpr int RCH_ROW = 32 * EB / 16;
  constexpr int TCH_ROW = KT * EB / 16;
  constexpr int CH = KT * RCH_ROW;            // 16-B chunks per tile: 256 (bf16) / 512 (f32)
  // staging: the 3 CH chunks of the K, V and Kt tiles plus the 64 KeyW records of the step are dealt over the
  // threads (1 chunk each in bf16 mode, 2 in f32 mode): per-thread source pointer (advanced by a fixed stride
  // per step) and LDS destination offset of each chunk
  constexpr int NCHUNK = 3 * CH + KT;
  constexpr int NCHK = (NCHUNK + TQ - 1) / TQ;
  u32x4 st[NCHK];
  const int n_step = d.Np / KT;
  // a chunk slot's tile is uniform over its wave (CH and KT are multiples of 64): uniform base pointer and step
  // stride, 32-bit per-thread byte offset
  const char* st_base[NCHK];
  unsigned st_off[NCHK];
  int st_inc[NCHK], st_dst[NCHK];
#pragma unroll
  for (int k = 0; k < NCHK; ++k) {
    const int g = tid + k * TQ;
    const int kind = __builtin_amdgcn_readfirstlane(g / CH), ci = g % CH;
    if (kind < 2) {
      st_base[k] = kind ? Vh : Kh;
      st_inc[k] = CH * 16;
      st_off[k] = (unsigned)ci * 16;
      st_dst[k] = kind * L::R_BYTES + (ci / RCH_ROW) * L::R_STRIDE + (ci % RCH_ROW) * 16;
    } else if (kind == 2) {
      st_base[k] = Kth;
      st_inc[k] = KT * EB;
      st_off[k] = (unsigned)(((size_t)(ci / TCH_ROW) * d.Np) * EB + (ci % TCH_ROW) * 16);
      st_dst[k] = 2 * L::R_BYTES + L::T_BYTES * 0 + (ci / TCH_ROW) * L::T_STRIDE + (ci % TCH_ROW) * 16;
    } else {
      st_base[k] = reinterpret_cast<const char*>(kws);
      st_inc[k] = KT * 16;
      st_off[k] = (unsigned)min(ci, KT - 1) * 16;
      st_dst[k] = g < NCHUNK ? 2 * L::R_BYTES + L::T_BYTES + ci * 16 : -1;   // past the last chunk: idle
    }
  }

  auto stage_load = [&](int step) {
#pragma unroll
    for (int k = 0; k < NCHK; ++k) {
      unsigned o = st_off[k];
      asm volatile("" : "+v"(o));   // keep the per-thread part a 32-bit offset (no hoisted 64-bit pointer to spill)
      if (st_dst[k] >= 0) st[k] = *reinterpret_cast<const u32x4*>(st_base[k] + (size_t)step * st_inc[k] + o);
    }
  };
  auto stage_store = [&]() {
    char* base = smem;
#pragma unroll
    for (int k = 0; k < NCHK; ++k)
      if (st_dst[k] >= 0) *reinterpret_cast<u32x4*>(base + st_dst[k]) = st[k];
  };

  stage_load(0);
  // the two kill columns of the table window; their accumulation cells only ever get +0.  The accumulation window
  // starts clean.
  for (int c = tid; c < L::WCOLS * WIN_PITCH; c += TQ) accw[c] = 0;
  if (tid < 2 * WIN_PITCH) {
    if constexpr (is16(PREC))
      *reinterpret_cast<unsigned*>(win + (CAP * WIN_PITCH + tid) * ENT) = Half<PREC>::pack2(Half<PREC>::NEG_BIG, Half<PREC>::NEG_BIG);
    else
      *reinterpret_cast<f32x2*>(win + (CAP * WIN_PITCH + tid) * ENT) = f32x2{BEVR_NEG_BIG, BEVR_NEG_BIG};
  }
  __syncthreads();

  Region rg;
  rg.ax0 = -(1 << 28);
  rg.ay0 = 0;
  // box of the region (region coordinates) that has received gradient since the last flush; uniform.  c1 < c0: clean.
  int dc0 = 1 << 20, dc1 = -1, dr0 = 1 << 20, dr1 = -1;

  // drain the dirty box of the shared window of region `r` (all threads): fixed point -> float, non-zero cells only,
  // then clear.  A column of the window is 64 consecutive rows of one table column: contiguous float atomics.
  auto flush_and_clear = [&](const Region& r) {
    if (dc1 >= dc0) {
      const int nr = dr1 - dr0 + 1, ncell = nr * (dc1 - dc0 + 1);
      const float inv_nr = 1.0f / (float)nr;
      for (int u = tid; u < ncell; u += TQ) {
        int c = (int)((float)u * inv_nr);          // u / nr, corrected for the float estimate
        int row = u - c * nr;
        if (row < 0) { row += nr; --c; }
        if (row >= nr) { row -= nr; ++c; }
        const int cell = (dc0 + c) * WIN_PITCH + dr0 + row;
        const acc_t v = accw[cell];
        if (v != 0) {
          accw[cell] = 0;
          BEVR_ASSERT(r.ax0 + dc0 + c + d.x_off < d.Wp);   // only columns of the real padded table ever receive gradient
          const float f = Acc::to_float(v) * ginv;
          atomicAdd(dtb + (size_t)(r.ax0 + dc0 + c + d.x_off) * Hq + (size_t)(i0 + r.ay0 + d.y_off + dr0 + row), f);
        }
      }
    }
    dc0 = 1 << 20; dc1 = -1; dr0 = 1 << 20; dr1 = -1;
  };

#ifdef BEVR_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // boxes of the two 32-key halves of a step: uniform (scalar loads), fetched one step ahead
  StepBox sb_cur[2] = {kbox[0], kbox[1]};
  StepBox sb_nxt[2] = {kbox[2 * min(1, n_step - 1)], kbox[2 * min(1, n_step - 1) + 1]};
  for (int step = 0; step < n_step; ++step) {
    PROF_T(t0);
    // single staging buffer: the registers hold this step's tiles (loaded during the previous step); every wave
    // finished reading the previous tiles at the barrier that ended the previous step
    const char* base = smem;
    stage_store();
    __syncthreads();
    if (step + 1 < n_step) stage_load(step + 1);
    const StepBox sb0 = sb_cur[0], sb1 = sb_cur[1];
    sb_cur[0] = sb_nxt[0];
    sb_cur[1] = sb_nxt[1];
    sb_nxt[0] = kbox[2 * min(step + 2, n_step - 1)];
    sb_nxt[1] = kbox[2 * min(step + 2, n_step - 1) + 1];
    // if the whole step's box fits, both halves share one region test (fewer moves); else the halves go separately
    const WinInfo wi_step = make_wininfo(box_union(sb0, sb1), jrx_lo, jrx_hi, CAP);
    PROF_T(t1);
    PROF_ADD(3, t1 - t0);   // staging store + barrier + box bookkeeping
    const KeyW* kc0 = reinterpret_cast<const KeyW*>(base + 2 * L::R_BYTES + L::T_BYTES);

#pragma unroll 1
    for (int kh = 0; kh < 2; ++kh) {   // the two 32-key halves of the step, one after the other
      const KeyW* kc = kc0 + kh * 32;
      const StepBox sbh = kh ? sb1 : sb0;
      const bool last = (step == n_step - 1) && d.N < d.Np;
      // How this half is served (all workgroup-uniform: functions of the scalar-loaded boxes only):
      //   its own box (or the whole step's) fits a region  -> one windowed pass over all 32 keys      (n_pass = 1)
      //   else, the half has groups (attn_tile.h)          -> one windowed pass per non-empty group
      //   else                                             -> the global-memory path                  (n_pass = 0)
      const WinInfo wi_half = wi_step.ok ? wi_step : make_wininfo(sbh, jrx_lo, jrx_hi, CAP);
      const StepBox* gb = gbox + (size_t)(2 * step + kh) * N_GROUP;
      const bool grouped = !wi_half.ok && sbh.amax >= sbh.amin && gb[0].amin != GROUPS_NONE;
      const int n_pass = wi_half.ok ? 1 : (grouped ? N_GROUP : 0);
      bool moved_in_half = false;

#pragma unroll 1
      for (int pass = 0; pass < n_pass; ++pass) {
        WinInfo wi = wi_half;
        int gsel = -1;
        if (grouped) {
          const StepBox gbx = gb[pass];
          if (gbx.amax < gbx.amin) continue;        // empty group (uniform)
          wi = make_wininfo(gbx, jrx_lo, jrx_hi, CAP);
          gsel = pass;
          BEVR_ASSERT(wi.ok);
        }
        BEVR_ASSERT_WG_UNIFORM(wi.xlo * 131 + wi.amin * 7 + wi.ncols + gsel * 977);
        PROF_T(tp);
        // ---- this pass's table window ------------------------------------------------------------------
        if (!region_contains(rg, wi, CAP)) {
          // every wave must be done with the taps and adds of the previous half / pass of this step
          if (kh || pass || moved_in_half) __syncthreads();
          flush_and_clear(rg);
          rg = region_anchor(wi, d, i0, CAP);
          {   // fill the region: one wave-wide load per table column (lane = row); all of a wave's loads are issued
              // before the first LDS store (one L2 latency per move instead of one per column)
            const size_t y0 = (size_t)(i0 + rg.ay0 + d.y_off) + lane;
            BEVR_ASSERT(i0 + rg.ay0 + d.y_off >= 0 && i0 + rg.ay0 + d.y_off + WIN_PITCH <= d.Hp && rg.ax0 + d.x_off >= 0);
            constexpr int PER_WAVE = (CAP + NWAVE - 1) / NWAVE;
            f32x2 fv[PER_WAVE];
#pragma unroll
            for (int k = 0; k < PER_WAVE; ++k) {
              const int c = wave + k * NWAVE;
              fv[k] = c < CAP ? region_entry(tbl, d, rg, c, y0) : f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < PER_WAVE; ++k) {
              const int c = wave + k * NWAVE;
              if (c < CAP) {
                if constexpr (is16(PREC))
                  *reinterpret_cast<unsigned*>(win + (c * WIN_PITCH + lane) * ENT) = Half<PREC>::pack2(fv[k][0], fv[k][1]);
                else
                  *reinterpret_cast<f32x2*>(win + (c * WIN_PITCH + lane) * ENT) = fv[k];
              }
            }
          }
          moved_in_half = true;
          __syncthreads();
          PROF_ADD(5, 1);
        }
        PROF_T(tq);
        PROF_ADD(0, tq - tp);   // region handling (moves: flush + refill + barriers)
        {   // the pass's cells become dirty (region coordinates)
          dc0 = min(dc0, wi.xlo - rg.ax0);
          dc1 = max(dc1, wi.xlo - rg.ax0 + wi.ncols - 1);
          dr0 = min(dr0, wi.amin - rg.ay0);
          dr1 = max(dr1, wi.amin - rg.ay0 + wi.nrows);
        }
        {
          // this wave's (column, key) constants for the 32 keys of this half: lane & 31 = key (both lane halves write
          // the same values); read back by this wave only -- a wave's LDS operations execute in order
          const KeyW kw = kc[lq];
          const float tx = jrx + (kw.b - (float)rg.ax0);
          const float xf = floorf(tx);
          CK e;
          // masked key (padding, or not of this pass's group): taps in the kill column => P = 0, dS = 0
          const bool dead = step * KT + kh * 32 + lq >= d.N || (gsel >= 0 && (kw.arow8 & 7) != gsel);
          const float fx = tx - xf, fy = kw.fy;
          if (dead) e.set(1.f, 0.f, 0.f, 0.f);
          else e.set((1.0f - fx) * (1.0f - fy), (1.0f - fx) * fy, fx * (1.0f - fy), fx * fy);
          e.cell = dead ? CAP * WIN_PITCH : (int)xf * WIN_PITCH + (kw.arow8 >> 3) + (sbh.amin - rg.ay0);   // arow8: relative to the half's first row
          BEVR_ASSERT(dead || (e.cell >= 0 && e.cell + 32 + WIN_PITCH < L::WCOLS * WIN_PITCH));
          pck[lq] = e;
        }
        // operand fragments are loaded one MFMA chain at a time (K with Q, then V with dO; K^T only after the loop):
        // all three at once put the kernel 16 registers over its 128 and the Q fragment went to scratch
        f32x16 s, dp;
        {
          // launder the row constants: otherwise the splatted 16-register accumulator seeds are hoisted out of the
          // step loop and live in scratch (reloaded every step, a full-latency miss each time)
          float nl = kp16 - lse, nd = -dlt;
          asm volatile("" : "+v"(nl), "+v"(nd));
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = nl; dp[r] = nd; }
        }
        {
          Frag<PREC> kf;
          kf.load(base + (kh * 32 + lq) * L::R_STRIDE, hi);
          s = mma_frag(kf, qf, s);        // S^T - LSE
        }
        {
          Frag<PREC> vkf;
          vkf.load(base + L::R_BYTES + (kh * 32 + lq) * L::R_STRIDE, hi);
          if constexpr (is16(PREC)) {
            const u32x4* qd = reinterpret_cast<const u32x4*>(qdo) + col * 2 * 64 + lane;
            Frag<PREC> dos;
            dos.v[0] = __builtin_bit_cast(bf16x8, qd[0]);
            dos.v[1] = __builtin_bit_cast(bf16x8, qd[64]);
            dp = mma_frag(vkf, dos, dp);    // dP^T - delta
          } else {
            dp = mma_frag(vkf, dof, dp);
          }
        }
#if BEVR_DROP
        {
          // dropout: O = sum_n D_n P_n V_n, D = keep / (1 - p)  =>  dS = P (D dP - delta); dp holds dP - delta
          const uint32_t hrow = bevr_drop_row(drop_seed, (uint32_t)ph, (uint32_t)mq);
          const float ksc = 65536.0f / (65536.0f - (float)drop_thr), nd2 = -dlt;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dp[r] = bevr_drop_keep(hrow, (uint32_t)(step * KT + kh * 32 + crow(r, hi)), drop_thr) ? fmaf(ksc, dp[r] - nd2, nd2) : nd2;
        }
#endif
        PROF_TD(t2, s[0] + dp[15]);
        PROF_ADD(1, t2 - tq);

        {
          // The LDS atomics are ordered memory operations for the compiler: it moves no load across them.  So
          // the loop is software-pipelined by hand -- the taps of key r + 1 and the constants of key r + 2 are
          // requested before the atomics of key r are issued, and their latency hides behind key r's arithmetic.
          typedef typename std::conditional<is16(PREC), unsigned, f32x2>::type tap_t;
          auto read_tap = [&](int cell, tap_t& a, tap_t& b) {
            const char* p = win + (cell + lq) * ENT;
            a = *reinterpret_cast<const tap_t*>(p);
            b = *reinterpret_cast<const tap_t*>(p + WIN_PITCH * ENT);
          };
          const CK* pk = pck;
          CK e0 = pk[crow(0, hi)], e1 = pk[crow(1, hi)];
          tap_t ta, tb;
          read_tap(e0.cell, ta, tb);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            tap_t na = ta, nb = tb;
            CK e2 = e1;
            if (r + 1 < 16) read_tap(e1.cell, na, nb);
            if (r + 2 < 16) e2 = pk[crow(r + 2, hi)];
            float sv;
            if constexpr (is16(PREC)) {
              sv = Half<PREC>::dot2(ta, e0.wA, s[r]);
              sv = Half<PREC>::dot2(tb, e0.wB, sv);
            } else {
              sv = fmaf(tb[1], e0.w11(), fmaf(tb[0], e0.w10(), fmaf(ta[1], e0.w01(), fmaf(ta[0], e0.w00(), s[r]))));
            }
            float ds = fast_exp2(sv) * dp[r];   // masked keys: a huge negative tap from the kill column => 0
            if constexpr (PREC == BEVR_PREC_F16) ds *= c2_16;   // P' (dP - delta) c2: inside fp16's normal range
            s[r] = ds;
            // table gradient.  Table row (A + l) of column X receives  w00 dS[query l] + w01 dS[query l - 1]  (the
            // upper tap of query l and the lower tap of the query above it); lane l adds exactly that, lane 31 the
            // lower tap of query 30 alone (its own dS is 0), lane 0 / lane 32 get 0 from below (wave_shr zero fill /
            // lane 31's zero).  One 64-bit fixed-point add per table column: order-free, bit-reproducible.
            const float g = ds;   // already in cell units (dO and delta carry the scale)
            const float gb_ = lane_below(g);
            acc_t* gp = accw + (e0.cell + lq);
            if constexpr (PREC == BEVR_PREC_BF16) {   // the weights the bias was computed with; dS in bf16 as for dQ
              // Both dot products and both round-to-integer conversions as ONE asm block: the VOP3P dot product takes
              // its zero addend inline (the compiler only selects v_dot2c, which needs a zeroed register first) and
              // v_cvt_rpi_i32_f32 = floor(x + 0.5) rounds and converts in one instruction -- 4 VALU instead of 8 in a
              // VALU-bound loop.  The s_nop is the dot -> dependent-VALU wait the hazard recognizer cannot insert for
              // instructions it does not see (without it the conversion read stale data).
              const unsigned pr = pack_bf16x2(g, gb_);
              int iA, iB;
              asm("v_dot2_f32_bf16 %0, %2, %3, 0\n\t"
                  "v_dot2_f32_bf16 %1, %2, %4, 0\n\t"
                  "s_nop 2\n\t"
                  "v_cvt_rpi_i32_f32 %0, %0\n\t"
                  "v_cvt_rpi_i32_f32 %1, %1"
                  : "=&v"(iA), "=&v"(iB)
                  : "v"(pr), "v"(e0.wA), "v"(e0.wB));
              atomicAdd(gp, Acc::from_int(iA));
              atomicAdd(gp + WIN_PITCH, Acc::from_int(iB));
            } else if constexpr (PREC == BEVR_PREC_F16) {
              // the same with fp16 operands; the dot products come out in dS16 units and are brought to cell units
              // (x cfix = 2^16) in f32 before the round-to-integer
              const unsigned pr = Half<PREC>::pack2(g, gb_);
              int iA, iB;
              asm("v_dot2_f32_f16 %0, %2, %3, 0\n\t"
                  "v_dot2_f32_f16 %1, %2, %4, 0\n\t"
                  "s_nop 2\n\t"
                  "v_mul_f32 %0, %0, %5\n\t"
                  "v_mul_f32 %1, %1, %5\n\t"
                  "v_cvt_rpi_i32_f32 %0, %0\n\t"
                  "v_cvt_rpi_i32_f32 %1, %1"
                  : "=&v"(iA), "=&v"(iB)
                  : "v"(pr), "v"(e0.wA), "v"(e0.wB), "v"(cfix));
              atomicAdd(gp, Acc::from_int(iA));
              atomicAdd(gp + WIN_PITCH, Acc::from_int(iB));
            } else {
              const float hA = fmaf(gb_, e0.w01(), g * e0.w00());
              const float hB = fmaf(gb_, e0.w11(), g * e0.w10());
              atomicAdd(gp, Acc::from(hA));
              atomicAdd(gp + WIN_PITCH, Acc::from(hB));
            }
            e0 = e1; e1 = e2; ta = na; tb = nb;
          }
        }
        PROF_TD(t3, s[15]);
        PROF_ADD(2, t3 - t2);
        {
          Frag<PREC> ktf;
          load_perm(ktf, base + 2 * L::R_BYTES + lq * L::T_STRIDE + kh * 32 * EB, hi);
          dq = mma_acc_b(ktf, s, dq);
        }
        PROF_TD(t3b, dq[0]);
        PROF_ADD(6, t3b - t3);
      }

      if (n_pass == 0 && sbh.amax >= sbh.amin) {
        // ---- global-memory path: a half spread over more table than its groups can band (or a table so wide that
        // no group fits a region).  Per-pair gathers from L2 and float atomics to HBM.
        Frag<PREC> kf, vkf, ktf;
        kf.load(base + (kh * 32 + lq) * L::R_STRIDE, hi);
        vkf.load(base + L::R_BYTES + (kh * 32 + lq) * L::R_STRIDE, hi);
        load_perm(ktf, base + 2 * L::R_BYTES + lq * L::T_STRIDE + kh * 32 * EB, hi);
        f32x16 s, dp;
        {
          float nl = kp16 - lse, nd = -dlt;
          asm volatile("" : "+v"(nl), "+v"(nd));
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = nl; dp[r] = nd; }
        }
        if constexpr (is16(PREC)) {
          const u32x4* qd = reinterpret_cast<const u32x4*>(qdo) + col * 2 * 64 + lane;
          Frag<PREC> dos;
          dos.v[0] = __builtin_bit_cast(bf16x8, qd[0]);
          dos.v[1] = __builtin_bit_cast(bf16x8, qd[64]);
          s = mma_frag(kf, qf, s);
          dp = mma_frag(vkf, dos, dp);
        } else {
          s = mma_frag(kf, qf, s);
          dp = mma_frag(vkf, dof, dp);
        }
#if BEVR_DROP
        {
          // dropout: O = sum_n D_n P_n V_n, D = keep / (1 - p)  =>  dS = P (D dP - delta); dp holds dP - delta
          const uint32_t hrow = bevr_drop_row(drop_seed, (uint32_t)ph, (uint32_t)mq);
          const float ksc = 65536.0f / (65536.0f - (float)drop_thr), nd2 = -dlt;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dp[r] = bevr_drop_keep(hrow, (uint32_t)(step * KT + kh * 32 + crow(r, hi)), drop_thr) ? fmaf(ksc, dp[r] - nd2, nd2) : nd2;
        }
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const KeyW c = kc[crow(r, hi)];
          const float wy0 = 1.0f - c.fy;
          float tx = jrx + c.b;
          float xf = floorf(tx);
          float fx = tx - xf;
          int xi = (int)xf;
          unsigned off = (unsigned)(xi * Hp8 + c.aoff + rowoff);
          f32x2 t0 = *reinterpret_cast<const f32x2*>(tbl + off);
          f32x2 t1 = *reinterpret_cast<const f32x2*>(tbl + off + Hp8);
          float u0 = t0[0] * wy0 + t0[1] * c.fy;
          float u1 = t1[0] * wy0 + t1[1] * c.fy;
          float sv = s[r] + u0 + fx * (u1 - u0);
          if (last && step * KT + kh * 32 + crow(r, hi) >= d.N) sv = BEVR_NEG_BIG;
          float ds = fast_exp2(sv) * dp[r];
          if constexpr (PREC == BEVR_PREC_F16) ds *= c2_16;
          s[r] = ds;
          if (ds != 0.f) {
            // plain transposed table, row pitch Hp + 1
            int yi = (c.aoff >> 3) - xoffHp + ilane;
            float* g0 = dtb + (size_t)(xi + d.x_off) * Hq + yi;
            float w0 = dq_scale * ds * (1.0f - fx), w1 = dq_scale * ds * fx;   // ln2 / (the scale ds carries)
            atomicAdd(g0, w0 * wy0);
            atomicAdd(g0 + 1, w0 * c.fy);
            atomicAdd(g0 + Hq, w1 * wy0);
            atomicAdd(g0 + Hq + 1, w1 * c.fy);
          }
        }
        dq = mma_acc_b(ktf, s, dq);
      }
    }

    PROF_T(t4);
    __syncthreads();   // every wave is done with the staged tiles (and with the region, should the next step move it)
    PROF_T(t5);
    PROF_ADD(4, t5 - t4);
    PROF_ADD(7, 1);
  }
#ifdef BEVR_PROF
  if (lane == 0 && (wave == 0 || wave == NWAVE - 1)) {
    for (int i = 0; i < 8; ++i) atomicAdd(&bevr_prof[(wave ? 8 : 0) + i], pacc[i]);
  }
#endif
  flush_and_clear(rg);

  // ---- store dQ (ln2 of dS = ln2 P (dP - delta) and the 2^-e of the cell scale applied here) -----------------------------------------------
  if (live) {
    float* row = dQ + ((size_t)ph * Mp + (size_t)jcol * d.Sp + qrow) * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = dq_scale * dq[4 * g4 + k];
      *reinterpret_cast<f32x4*>(row + 8 * g4 + 4 * hi) = v;
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* K, const void* Kt, const void* V, const void* key_ws,
           const float* table_pair, const void* dO, const float* LSE, const float* delta,
           const float* grad_scale, float* dQ, float* dtable, hipStream_t st BEVR_DROP_PARAMS) {
  const int n_rb = (d.S + QROWS - 1) / QROWS, n_cb = (d.S + NCOL - 1) / NCOL;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_rb * n_cb;
  hipLaunchKernelGGL((attn_bwd_q_kernel<PREC>), dim3(grid), dim3(TQ), 0, st, d, (const char*)Q, (const char*)K,
                     (const char*)Kt, (const char*)V, (const char*)key_ws, (const char*)table_pair, (const char*)dO, LSE,
                     delta, grad_scale, dQ, dtable BEVR_DROP_ARGS);
  return (int)hipGetLastError();
}

}  // namespace

#if BEVR_DROP
extern "C" int bevr_attn_bwd_q_dropout(const bevr_attn_desc* d, const void* Q, const void* K, const void* Kt, const void* V,
                                       const void* key_ws, const float* table_pair, const void* dO,
                                       const float* LSE, const float* delta, const float* grad_scale, float* dQ,
                                       float* dtable, unsigned drop_thr, unsigned drop_seed, void* stream) {
  if (drop_thr >= 65536u) return BEVR_E_SHAPE;
#else
extern "C" int bevr_attn_bwd_q(const bevr_attn_desc* d, const void* Q, const void* K, const void* Kt, const void* V,
                               const void* key_ws, const float* table_pair, const void* dO,
                               const float* LSE, const float* delta, const float* grad_scale, float* dQ,
                               float* dtable, void* stream) {
#endif
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !K || !Kt || !V || !key_ws || !table_pair || !dO || !LSE || !delta || !grad_scale || !dQ ||
      !dtable)
    return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(K) || !bevr_aligned16(Kt) || !bevr_aligned16(V) || !bevr_aligned16(dO) ||
      !bevr_aligned16(dQ) || !bevr_aligned16(table_pair))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable,
                                  st BEVR_DROP_ARGS);
  if (d->precision == BEVR_PREC_F16)
    return launch<BEVR_PREC_F16>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st BEVR_DROP_ARGS);
  if (d->precision == BEVR_PREC_BF16X3)
    return launch<BEVR_PREC_BF16X3>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st BEVR_DROP_ARGS);
  return launch<BEVR_PREC_F32>(*d, Q, K, Kt, V, key_ws, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st BEVR_DROP_ARGS);
}
