// Attention forward over a CELL-SORTED key segment (attn_cell.h): O = softmax_n(Q^T K scale + rpe bias) V with the
// bias of a (32 keys x 32 BEV rows) tile as ONE extra MFMA.  Same arithmetic as attn_fwd.hip
// (model/SCA_deform_attn.py:331-413 of the reference), same operand layouts, same orientation (S^T[key][query],
// query on the lane, online softmax, P^T fed back as the B operand of PV).
//
// Work split: workgroup = ONE BEV column j of one (problem, head): one wave per 32-row block of the column + one
// PRODUCER wave.  Every row-block wave needs the same per-(column, key) weights W, the same tile geometry and the same
// staged K / V^T tiles: the producer fetches the next step's keys (global -> registers -> LDS) and builds W and the
// geometry for it while the row-block waves compute the current step; one barrier per 64-key step hands the buffer over.
// (With the staging shared by all waves and a rotating builder wave the kernel was 25 % slower: every wave carried
// staging registers and instructions, and the builder arrived last at every barrier.)  A row-block wave reloads its
// table operand only when the chunk origin changes (3 % of the tiles of a cell-sorted segment).
//
// The kernel is VALU-bound (a wave64 VALU instruction holds the SIMD ~4.4 clk; measured, profiles/r03_pmc_*), so the
// tile body is written for instruction count: the MFMA chains start from a literal-zero accumulator and the row
// constant is subtracted with packed adds; there is no running-max chain -- P = exp2(S - m) is formed against the
// reference m as it stands, the tile is committed, and m is moved up AFTERWARDS when the tile's mass exceeds 4 (so every
// committed weight is <= 4 relative to the final m, as in attn_fwd.hip); only a tile whose mass overflows (logits
// more than ~60 above m: the first tile, or an adversarial jump) is redone with its exact maximum.
//
// Two passes (template parameter SLOW): the fast pass computes the tiles that fit one chunk and skips the others; the slow
// pass lists the skipped tiles of its column (its workgroups exit at once when there is none: the normal case for a
// cell-sorted segment), stages just their steps and computes them with a per-pair gather from the table in global
// memory, continuing the softmax in place from the fast pass's (O, LSE).
//
// Chaining: the keys of one softmax may be split between the region kernels (scattered keys) and this one.  With
// (O_in, LSE_in) given, the online softmax starts from that state (m = LSE_in, l = 1, o = O_in) and the result is the
// softmax over both segments; with O_in == NULL it starts empty.
#include <type_traits>
#include "attn_cell.h"


namespace {

constexpr float MASS_THR = 4.0f;      // move the reference up when a tile's mass exceeds this: every weight <= 4 = 2^2
constexpr float LOG2_MASS_THR = 2.0f; // ... which is also the slack of LSE plane 1 (as attn_fwd.hip's RESCALE_THR)
constexpr float MASS_REDO = 1.0e18f;  // a tile this heavy (or inf / NaN) is redone with its exact maximum
// fp16 operands: the reference sits Half::SHIFT = 10 binades under the maximum (attn_fwd.hip), weights may reach 2^12
// before the reference moves, and a tile whose mass nears fp16's largest number (65504) is redone exactly
template <int PREC> constexpr float shift16() { return PREC == BEVR_PREC_F16 ? 10.0f : 0.0f; }
template <int PREC> constexpr float mass_thr() { return PREC == BEVR_PREC_F16 ? MASS_THR * 1024.0f : MASS_THR; }
template <int PREC> constexpr float mass_redo() { return PREC == BEVR_PREC_F16 ? 30000.0f : MASS_REDO; }
constexpr float CHAIN_SHIFT16 = 12.0f;   // fp16: an incoming (O, LSE) state is re-referenced 12 binades down (l = 2^12)

template <int PREC> struct LdsC {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int K_STRIDE = 32 * EB + 16;
  static constexpr int V_STRIDE = KT * EB + 16;
  static constexpr int K_BYTES = KT * K_STRIDE;
  static constexpr int V_BYTES = 32 * V_STRIDE;
  static constexpr int KW_BYTES = KT * 16;
  static constexpr int WL = 8 * EB;                 // bytes of one lane's chunk operand
  static constexpr int W_BYTES = 2 * 64 * WL;       // two tiles per step
  static constexpr int CT_BYTES = 2 * 16;           // two CellTile records
  static constexpr int OFF_V = K_BYTES, OFF_KW = K_BYTES + V_BYTES, OFF_W = OFF_KW + KW_BYTES, OFF_CT = OFF_W + W_BYTES;
  static constexpr int OFF_DUMMY = OFF_CT + CT_BYTES;   // 16 B that idle staging threads write (keeps staging branch-free)
  static constexpr int BUF = OFF_DUMMY + 16;
  static constexpr int KCH_ROW = 32 * EB / 16;      // 16-B chunks per K row
  static constexpr int VCH_ROW = KT * EB / 16;      // 16-B chunks per V^T row of this step
  static constexpr int CH = KT * KCH_ROW;           // chunks per tile (K rows, V^T rows): 256 / 512
  static constexpr int NCH = 2 * CH + KT;           // + one KeyW record per key
  static constexpr int NST = is16(PREC) ? 2 : 3;   // chunks a thread carries in registers across a step
  static constexpr int QCH = 32 * EB / 16 / 2;      // 16-B chunks of one lane's Q fragment: 2 (bf16) / 4 (f32)
  static constexpr int QSLOT = QCH * 1024;          // per wave: [chunk][lane] -- consecutive lanes, consecutive 16 B
};

// staging chunk g of a step: source at step 0, byte increment per step, LDS destination inside a buffer
template <int PREC>
__device__ __forceinline__ void chunk_map(int g, const char* Kh, const char* Vh, const char* kws, int Np,
                                          const char*& src, int& inc, int& dst) {
  typedef LdsC<PREC> L;
  constexpr int EB = L::EB;
  if (g < L::CH) {
    src = Kh + (size_t)g * 16;
    inc = L::CH * 16;
    dst = (g / L::KCH_ROW) * L::K_STRIDE + (g % L::KCH_ROW) * 16;
  } else if (g < 2 * L::CH) {
    const int ci = g - L::CH;
    src = Vh + ((size_t)(ci / L::VCH_ROW) * Np) * EB + (ci % L::VCH_ROW) * 16;
    inc = KT * EB;
    dst = L::OFF_V + (ci / L::VCH_ROW) * L::V_STRIDE + (ci % L::VCH_ROW) * 16;
  } else {
    const int ci = g - 2 * L::CH;
    src = kws + (size_t)ci * 16;
    inc = KT * 16;
    dst = L::OFF_KW + ci * 16;
  }
}

// MAXT: the launch bound.  1024 threads (BEV sides up to 512) cap a wave at 128 registers; the f32-layout modes need
// more than that (fragments twice as wide) and get a 512-thread instantiation (BEV sides up to 256) without spills.
template <int PREC, bool SLOW, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_cell_fwd_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ Vt,
    const char* __restrict__ key_ws, const char* __restrict__ table_pair, const float* O_in,
    const float* LSE_in, float* O, float* LSE) {   // (O_in, LSE_in) may alias (O, LSE): the chained call passes them so
  typedef LdsC<PREC> L;
  constexpr int EB = L::EB;
  // 2 staging buffers | one Q fragment slot per wave | (slow pass) the list of this column's slow tiles
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
  // fast pass: the LAST wave is the workgroup's PRODUCER -- it stages the next step's keys and builds their weight
  // tiles while the other waves (one per 32-row block) run the tiles; it owns no BEV rows (its row indices alias row
  // block 0 so that the prologue below stays in range; it leaves before the epilogue)
  const bool producer = !SLOW && wave == n_wave - 1;
  const int i0 = (producer ? 0 : wave) * 32;

  const char* Qh = Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB;
  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = Vt + ((size_t)ph * 32) * d.Np * EB;
  const int pg = prob * d.groups + grp;
  const char* kws = key_ws + (size_t)pg * d.Np * sizeof(KeyW);
  const StepBox* kbox = reinterpret_cast<const StepBox*>(key_ws + key_ws_box_offset(d)) + (size_t)pg * (d.Np / 32);
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const int Hp8 = d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const float jrx = (float)j * rx;
  const size_t mq = (size_t)j * d.Sp + i0 + lq;
  const int n_step = d.Np / KT;

  char* qslot = smem + 2 * L::BUF + wave * L::QSLOT + lane * 16;
  int* slow_list = reinterpret_cast<int*>(smem + 2 * L::BUF + n_wave * L::QSLOT);
  __shared__ int slow_count;
  if constexpr (SLOW) {
    // this column's slow tiles, listed in key order (the order of the list is the summation order: deterministic).
    // One tile box per thread and round -> a flag byte; then wave 0 compacts the flags with ballots.
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

  // the Q fragment lives in LDS (own lanes' data, written and read by this wave only: no barrier) and is re-read per
  // tile: 8 (bf16) / 16 (f32) registers less to carry through the loop
  {
    Frag<PREC> qf;
    qf.load(Qh + mq * 32 * EB, hi);
    if constexpr (is16(PREC)) {
      *reinterpret_cast<u32x4*>(qslot) = __builtin_bit_cast(u32x4, qf.v[0]);
      *reinterpret_cast<u32x4*>(qslot + 1024) = __builtin_bit_cast(u32x4, qf.v[1]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(qslot + 1024 * k) = f32x4{qf.v[4 * k], qf.v[4 * k + 1], qf.v[4 * k + 2], qf.v[4 * k + 3]};
    }
  }
  auto load_q = [&](Frag<PREC>& f) {
    if constexpr (is16(PREC)) {
      f.v[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qslot));
      f.v[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qslot + 1024));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qslot + 1024 * k);
        f.v[4 * k] = t[0]; f.v[4 * k + 1] = t[1]; f.v[4 * k + 2] = t[2]; f.v[4 * k + 3] = t[3];
      }
    }
  };

  // ---- online-softmax state --------------------------------------------------------------------------
  f32x16 o;
  float m = 0.f, l = 0.f;
  bool first = true;   // the reference m is not set yet (wave-uniform)
  if (O_in) {
    const float lse_in = LSE_in[(size_t)ph * Mp + mq];
    // rows never written hold -inf in the incoming plane: start them from 0 (their results are never read)
    if (__any(lse_in > -3.0e38f)) {
      first = false;
      // fp16: weights relative to the incoming LSE would sit far below fp16's normal range; the same state re-referenced
      // CHAIN_SHIFT16 binades down (m - 12, l = 2^12, o = 2^12 O_in) is exact in f32 and keeps them representable
      constexpr float CS = PREC == BEVR_PREC_F16 ? CHAIN_SHIFT16 : 0.f;
      m = (lse_in > -3.0e38f ? lse_in : 0.f) - CS;
      l = hi == 0 ? __builtin_amdgcn_exp2f(CS) : 0.f;       // the halves' denominators are added in the epilogue
      const float* orow = O_in + ((size_t)ph * Mp + mq) * 32;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(orow + 8 * g4 + 4 * hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[4 * g4 + k] = v[k] * __builtin_amdgcn_exp2f(CS);
      }
    }
  }
  if (first) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
  }

  // ---- one tile of the slow pass: S^T = K Q^T + gathered bias, P = exp2(S^T - m), l += sum P, o += V^T P^T -----------
  // MASKED: the tile may hold padded keys (last step only).  A separate instantiation: inside one body the compiler
  // hoisted the 16 key-index compares out of the `last step` branch and every tile paid 32 instructions for them.
  auto tile = [&](auto masked_tag, const char* base, int step, int t) {
    constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll 1
    for (int attempt = 0; attempt < 2; ++attempt) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;   // literal-zero accumulator: no register splat
      {
        Frag<PREC> kf;
        kf.load(base + (t * 32 + lq) * L::K_STRIDE, hi);
        Frag<PREC> qf;
        load_q(qf);
        s = mma_frag(kf, qf, s);   // S^T[key][query]
      }
      {
        // per-pair gather from the table in global memory (any key set)
        const KeyW* kwl = reinterpret_cast<const KeyW*>(base + L::OFF_KW);
        const int rowoff = (i0 + lq) * 8;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const KeyW c = kwl[t * 32 + crow(r, hi)];
          const float wy0 = 1.0f - c.fy;
          const float tx = jrx + c.b;
          const float xf = floorf(tx);
          const float fx = tx - xf;
          const unsigned off = (unsigned)((int)xf * Hp8 + c.aoff + rowoff);
          const f32x2 t0 = *reinterpret_cast<const f32x2*>(tbl + off);
          const f32x2 t1 = *reinterpret_cast<const f32x2*>(tbl + off + Hp8);
          const float u0 = t0[0] * wy0 + t0[1] * c.fy;
          const float u1 = t1[0] * wy0 + t1[1] * c.fy;
          s[r] += u0 + fx * (u1 - u0);
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight (register budget)
        }
      }
      if constexpr (MASKED) {   // padded keys: no weight
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (step * KT + t * 32 + crow(r, hi) >= d.N) ? BEVR_NEG_BIG : s[r];
      }
      if (first || attempt == 1) {   // exact maximum of the tile: sets / moves the reference before the weights are formed
        float tm = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, s[r]);
        tm = fmaxf(tm, __shfl_xor(tm, 32)) - shift16<PREC>();   // the lane halves hold the same queries, different keys
        const float mn = first ? tm : fmaxf(m, tm);
        const float al = first ? 0.f : fast_exp2(m - mn);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= al;
        l *= al;
        m = mn;
        first = false;
      }
      const f32x2 nm = {-m, -m};
      f32x2 ls2 = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sh = f32x2{s[r], s[r + 1]} + nm;
        const f32x2 pp = {fast_exp2(sh[0]), fast_exp2(sh[1])};
        s[r] = pp[0];
        s[r + 1] = pp[1];
        ls2 += pp;
      }
      const float ts = ls2[0] + ls2[1];
      if (attempt == 0 && __any(!(ts <= mass_redo<PREC>()))) continue;   // overflowed against the old reference: redo exactly
      l += ts;
      {
        Frag<PREC> vf;
        load_perm(vf, base + L::OFF_V + lq * L::V_STRIDE + t * 32 * EB, hi);
        o = mma_acc_b(vf, s, o);
      }
      if (__any(ts > mass_thr<PREC>())) {   // wave-uniform, rare: keep every committed weight <= mass_thr of the reference
        const float tb = ts + __shfl_xor(ts, 32);
        const float up = fmaxf(ceilf(__log2f(tb)) - shift16<PREC>(), 0.f);
        const float al = fast_exp2(-up);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= al;
        l *= al;
        m += up;
      }
      break;
    }
  };


  // ---- the two tiles of a step of the fast pass in ONE straight-line body.  A wave parks ~2/3 of its time on the serial
  // chain LDS -> 3 MFMAs -> exp -> sum -> LDS -> 2 MFMAs of a tile (PMC: profiles/r03_pmc_*; exp, barrier, staging each
  // cost < 10 %); two independent chains let the second tile's MFMAs run under the first one's exponentials.  A tile
  // that is dead or left to the slow pass (ok = 0) gets weight 0.  One body for every case: a second, per-tile body for
  // the rare cases next to this one cost the loop 130 B of spills per lane (22 -> 43 ms).
  CellFrag<PREC> tf0, tf1;    // table operands of the chunks of tile 0 / 1, and their origins
  int tag_x0 = 1 << 30, tag_a0 = 1 << 30, tag_x1 = 1 << 30, tag_a1 = 1 << 30;
  if constexpr (is16(PREC)) {
    tf0.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    tf1.v = tf0.v;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) { tf0.v[k] = 0.f; tf1.v[k] = 0.f; }
  }
  auto load_w = [&](CellFrag<PREC>& wf, const char* wsrc) {
    if constexpr (is16(PREC)) {
      wf.v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wsrc));
    } else {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wsrc), w1 = *reinterpret_cast<const f32x4*>(wsrc + 16);
      wf.v[0] = w0[0]; wf.v[1] = w0[1]; wf.v[2] = w0[2]; wf.v[3] = w0[3];
      wf.v[4] = w1[0]; wf.v[5] = w1[1]; wf.v[6] = w1[2]; wf.v[7] = w1[3];
    }
  };
  auto tile2 = [&](auto masked_tag, const char* base, int step, int ok0, int ok1, int x00, int a00, int x01, int a01) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    if (ok0 && (x00 != tag_x0 || a00 != tag_a0)) {   // uniform: new chunk origin
      tf0 = cell_table<PREC>(tbl, d, x00, a00 + i0 + lq, hi);
      tag_x0 = x00;
      tag_a0 = a00;
    }
    if (ok1 && (x01 != tag_x1 || a01 != tag_a1)) {
      tf1 = cell_table<PREC>(tbl, d, x01, a01 + i0 + lq, hi);
      tag_x1 = x01;
      tag_a1 = a01;
    }
#pragma unroll 1
    for (int attempt = 0; attempt < 2; ++attempt) {
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }   // literal-zero accumulators: no register splat
      {
        Frag<PREC> k0, k1, qf;
        k0.load(base + lq * L::K_STRIDE, hi);
        k1.load(base + (32 + lq) * L::K_STRIDE, hi);
        load_q(qf);
        s0 = mma_frag(k0, qf, s0);   // S^T[key][query]
        s1 = mma_frag(k1, qf, s1);
      }
      {
        CellFrag<PREC> w0, w1;
        load_w(w0, base + L::OFF_W + lane * L::WL);
        load_w(w1, base + L::OFF_W + (64 + lane) * L::WL);
        s0 = mma_cell(w0, tf0, s0);   // + bias^T[key][query]
        s1 = mma_cell(w1, tf1, s1);
      }
      if (!ok0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = BEVR_NEG_BIG;
      }
      if (!ok1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[r] = BEVR_NEG_BIG;
      }
      if constexpr (MASKED) {   // padded keys: no weight
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] = (step * KT + crow(r, hi) >= d.N) ? BEVR_NEG_BIG : s0[r];
          s1[r] = (step * KT + 32 + crow(r, hi) >= d.N) ? BEVR_NEG_BIG : s1[r];
        }
      }
      if (first || attempt == 1) {   // exact maximum of the step: sets / moves the reference before the weights are formed
        float tm = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, fmaxf(s0[r], s1[r]));
        tm = fmaxf(tm, __shfl_xor(tm, 32)) - shift16<PREC>();   // the lane halves hold the same queries, different keys
        const float mn = first ? tm : fmaxf(m, tm);
        const float al = first ? 0.f : fast_exp2(m - mn);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= al;
        l *= al;
        m = mn;
        first = false;
      }
      const f32x2 nm = {-m, -m};
      f32x2 ls2 = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sh = f32x2{s0[r], s0[r + 1]} + nm;
        const f32x2 pp = {fast_exp2(sh[0]), fast_exp2(sh[1])};
        s0[r] = pp[0];
        s0[r + 1] = pp[1];
        ls2 += pp;
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sh = f32x2{s1[r], s1[r + 1]} + nm;
        const f32x2 pp = {fast_exp2(sh[0]), fast_exp2(sh[1])};
        s1[r] = pp[0];
        s1[r + 1] = pp[1];
        ls2 += pp;
      }
      const float ts = ls2[0] + ls2[1];
      if (attempt == 0 && __any(!(ts <= mass_redo<PREC>()))) continue;   // overflowed against the old reference: redo exactly
      l += ts;
      {
        Frag<PREC> v0, v1;
        load_perm(v0, base + L::OFF_V + lq * L::V_STRIDE, hi);
        load_perm(v1, base + L::OFF_V + lq * L::V_STRIDE + 32 * EB, hi);
        o = mma_acc_b(v0, s0, o);
        o = mma_acc_b(v1, s1, o);
      }
      if (__any(ts > mass_thr<PREC>())) {   // wave-uniform, rare: keep every committed weight <= mass_thr of the reference
        const float tb = ts + __shfl_xor(ts, 32);
        const float up = fmaxf(ceilf(__log2f(tb)) - shift16<PREC>(), 0.f);
        const float al = fast_exp2(-up);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= al;
        l *= al;
        m += up;
      }
      break;
    }
  };

  // f32-layout modes (fragments twice as wide): the same, ONE tile at a time -- the joint body needs 214 registers there,
  // this one 156 (512-thread instantiation, no spills): 48 -> 44 ms on tools/prof_cell.py in the split-bf16 mode.  At the
  // 1024-thread bound (128 registers, two workgroups per CU) it spills 88 B per lane and is slower (48.6 ms).  (In the 16-bit modes this second body next to the joint
  // one is what spilled; here it is the only one: a generic lambda is only instantiated where it is called.)
  auto tile1 = [&](auto masked_tag, auto t_tag, const char* base, int step, int x0, int a0) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr int T = decltype(t_tag)::value;
    CellFrag<PREC>& tf = T ? tf1 : tf0;
    int& tag_x = T ? tag_x1 : tag_x0;
    int& tag_a = T ? tag_a1 : tag_a0;
    if (x0 != tag_x || a0 != tag_a) {   // uniform: new chunk origin
      tf = cell_table<PREC>(tbl, d, x0, a0 + i0 + lq, hi);
      tag_x = x0;
      tag_a = a0;
    }
#pragma unroll 1
    for (int attempt = 0; attempt < 2; ++attempt) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      {
        Frag<PREC> kf, qf;
        kf.load(base + (T * 32 + lq) * L::K_STRIDE, hi);
        load_q(qf);
        s = mma_frag(kf, qf, s);   // S^T[key][query]
      }
      {
        CellFrag<PREC> wf;
        load_w(wf, base + L::OFF_W + (T * 64 + lane) * L::WL);
        s = mma_cell(wf, tf, s);   // + bias^T[key][query]
      }
      if constexpr (MASKED) {   // padded keys: no weight
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (step * KT + T * 32 + crow(r, hi) >= d.N) ? BEVR_NEG_BIG : s[r];
      }
      if (first || attempt == 1) {   // exact maximum of the tile: sets / moves the reference before the weights are formed
        float tm = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, s[r]);
        tm = fmaxf(tm, __shfl_xor(tm, 32)) - shift16<PREC>();
        const float mn = first ? tm : fmaxf(m, tm);
        const float al = first ? 0.f : fast_exp2(m - mn);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= al;
        l *= al;
        m = mn;
        first = false;
      }
      const f32x2 nm = {-m, -m};
      f32x2 ls2 = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sh = f32x2{s[r], s[r + 1]} + nm;
        const f32x2 pp = {fast_exp2(sh[0]), fast_exp2(sh[1])};
        s[r] = pp[0];
        s[r + 1] = pp[1];
        ls2 += pp;
      }
      const float ts = ls2[0] + ls2[1];
      if (attempt == 0 && __any(!(ts <= mass_redo<PREC>()))) continue;   // overflowed against the old reference: redo exactly
      l += ts;
      {
        Frag<PREC> vf;
        load_perm(vf, base + L::OFF_V + lq * L::V_STRIDE + T * 32 * EB, hi);
        o = mma_acc_b(vf, s, o);
      }
      if (__any(ts > mass_thr<PREC>())) {   // wave-uniform, rare: keep every committed weight <= mass_thr of the reference
        const float tb = ts + __shfl_xor(ts, 32);
        const float up = fmaxf(ceilf(__log2f(tb)) - shift16<PREC>(), 0.f);
        const float al = fast_exp2(-up);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= al;
        l *= al;
        m += up;
      }
      break;
    }
  };
  // both tiles of a step
  auto tiles = [&](auto masked_tag, const char* base, int step, int ok0, int ok1, int x00, int a00, int x01, int a01) {
    if constexpr (is16(PREC)) {
      tile2(masked_tag, base, step, ok0, ok1, x00, a00, x01, a01);
    } else {
      if (ok0) tile1(masked_tag, std::integral_constant<int, 0>{}, base, step, x00, a00);
      if (ok1) tile1(masked_tag, std::integral_constant<int, 1>{}, base, step, x01, a01);
    }
  };

  // copy a whole step through (the slow pass: every thread takes its share of the chunks)
  auto stage_direct = [&](char* base, int step, int g0) {
    for (int g = g0; g < L::NCH; g += nt) {
      const char* src;
      int inc, dst;
      chunk_map<PREC>(g, Kh, Vh, kws, d.Np, src, inc, dst);
      *reinterpret_cast<u32x4*>(base + dst) = *reinterpret_cast<const u32x4*>(src + (size_t)step * inc);
    }
  };

  if constexpr (SLOW) {
    // ---- slow pass: only the listed tiles; their steps are staged on demand -----------------------------------
    const int n_slow = slow_count;
    for (int u = 0; u < n_slow; ++u) {
      const int tile_id = slow_list[u];
      const int step = tile_id >> 1, t = tile_id & 1;
      __syncthreads();                  // every wave is done with the previous tile's buffer
      stage_direct(smem, step, tid);
      __syncthreads();
      if (step == n_step - 1 && d.N < d.Np) tile(std::true_type{}, smem, step, t);
      else tile(std::false_type{}, smem, step, t);
    }
  } else {
    // ---- fast pass: pipelined over the steps --------------------------------------------------------------------
    // a last step with padded keys is peeled off behind the loop (its masked tile body inside the loop cost every tile
    // registers or hoisted compares)
    const int n_main = d.N < d.Np ? n_step - 1 : n_step;
    if (producer) {
      // ---- the producer wave: global -> registers -> LDS one step ahead, and the weight tiles / geometry of the step.
      // Until round 3's last day every wave carried a share of the staging (8 registers, 2 loads, 2 LDS stores per
      // step) and a rotating builder wave the weights: the builder arrived last at every barrier (probe: -9 %).
      constexpr int NSTP = (L::NCH + 63) / 64;        // 16-byte chunks per lane and step
      u32x4 st[NSTP];
      const char* st_src[NSTP];
      int st_inc[NSTP], st_dst[NSTP];
#pragma unroll
      for (int k = 0; k < NSTP; ++k) {
        const int g = lane + 64 * k;
        // lanes beyond the chunk count re-read chunk 0 into a dummy slot: loads and stores stay unconditional
        chunk_map<PREC>(g < L::NCH ? g : 0, Kh, Vh, kws, d.Np, st_src[k], st_inc[k], st_dst[k]);
        if (g >= L::NCH) st_dst[k] = L::OFF_DUMMY;
      }
      // weights and geometry of tile t of a step (lane & 31 = key), into buffer `buf`
      auto build_w = [&](int buf, int step, int t, const KeyW& kw, const StepBox& sb) {
        const CellTile ct = make_celltile(sb, jrx);
        float tcol, trow;
        cell_coords(kw, jrx, ct.x0, step * KT + t * 32 + lq >= d.N, tcol, trow);
        const CellFrag<PREC> w = cell_weights<PREC>(tcol, trow, hi);
        char* bb = smem + buf * L::BUF;
        char* dst = bb + L::OFF_W + (t * 64 + lane) * L::WL;
        if constexpr (is16(PREC)) {
          *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, w.v);
        } else {
          *reinterpret_cast<f32x4*>(dst) = f32x4{w.v[0], w.v[1], w.v[2], w.v[3]};
          *reinterpret_cast<f32x4*>(dst + 16) = f32x4{w.v[4], w.v[5], w.v[6], w.v[7]};
        }
        if (lane == 0) *reinterpret_cast<CellTile*>(bb + L::OFF_CT + t * 16) = ct;
      };
      auto load_kw = [&](int step, int t) {
        return *reinterpret_cast<const KeyW*>(kws + ((size_t)step * KT + t * 32 + lq) * sizeof(KeyW));
      };
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
      // the tiles' geometry, computed once by the producer: one broadcast read each instead of ~40 instructions per wave
      const u32x4 cw0 = *reinterpret_cast<const u32x4*>(base + L::OFF_CT);
      const u32x4 cw1 = *reinterpret_cast<const u32x4*>(base + L::OFF_CT + 16);
      const int ok0 = __builtin_amdgcn_readfirstlane((int)(cw0[0] & cw0[1])), ok1 = __builtin_amdgcn_readfirstlane((int)(cw1[0] & cw1[1]));
      const int x00 = __builtin_amdgcn_readfirstlane((int)cw0[2]), a00 = __builtin_amdgcn_readfirstlane((int)cw0[3]);
      const int x01 = __builtin_amdgcn_readfirstlane((int)cw1[2]), a01 = __builtin_amdgcn_readfirstlane((int)cw1[3]);
      if (ok0 | ok1) tiles(std::false_type{}, base, step, ok0, ok1, x00, a00, x01, a01);
      __syncthreads();
    }
    if (n_main < n_step) {   // the peeled last step: padded keys masked
      const char* base = smem + (n_main & 1) * L::BUF;
      const u32x4 cw0 = *reinterpret_cast<const u32x4*>(base + L::OFF_CT);
      const u32x4 cw1 = *reinterpret_cast<const u32x4*>(base + L::OFF_CT + 16);
      const int ok0 = __builtin_amdgcn_readfirstlane((int)(cw0[0] & cw0[1])), ok1 = __builtin_amdgcn_readfirstlane((int)(cw1[0] & cw1[1]));
      if (ok0 | ok1)
        tiles(std::true_type{}, base, n_main, ok0, ok1, __builtin_amdgcn_readfirstlane((int)cw0[2]),
              __builtin_amdgcn_readfirstlane((int)cw0[3]), __builtin_amdgcn_readfirstlane((int)cw1[2]),
              __builtin_amdgcn_readfirstlane((int)cw1[3]));
    }
  }

  // ---- epilogue: normalise, store O^T tile as [q][32] rows and the log2-sum-exp planes -------------------
  {
    float* Oh = O + ((size_t)ph * Mp) * 32;
    float* Lh = LSE + (size_t)ph * Mp;
    const float lt = l + __shfl_xor(l, 32);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    float* orow = Oh + mq * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = o[4 * g4 + k] * inv;
      *reinterpret_cast<f32x4*>(orow + 8 * g4 + 4 * hi) = v;
    }
    if (hi == 0) {
      Lh[mq] = m + __log2f(lt);
      // plane 1: upper bound of log2 of the row's largest softmax weight (every committed weight <= MASS_THR 2^m)
      Lh[(size_t)n_ph * Mp + mq] = LOG2_MASS_THR + shift16<PREC>() - __log2f(lt);
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* K, const void* Vt, const void* key_ws,
           const float* table_pair, const float* O_in, const float* LSE_in, float* O, float* LSE, hipStream_t st) {
  typedef LdsC<PREC> L;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * d.S;
  const int n_rb = d.Sp / 32;                 // one wave per 32-row block of the column ...
  const int n_fast = n_rb + 1;                // ... + the producer wave (fast pass)
  if (n_fast > 16) return BEVR_E_SHAPE;
  const size_t lds = 2 * L::BUF + (size_t)n_fast * L::QSLOT;
  // both passes' LDS before anything is launched: the fast pass updates (O, LSE) in place
  const size_t lds_slow = 2 * L::BUF + (size_t)n_rb * L::QSLOT + (size_t)(d.Np / 32) * 4;
  if (lds > 160 * 1024 || lds_slow > 160 * 1024) return BEVR_E_SHAPE;
  if (!is16(PREC) && 64 * n_fast <= 512)
    hipLaunchKernelGGL((attn_cell_fwd_kernel<PREC, false, (is16(PREC) ? 1024 : 512)>), dim3(grid), dim3(64 * n_fast), lds, st,
                       d, (const char*)Q, (const char*)K, (const char*)Vt, (const char*)key_ws, (const char*)table_pair,
                       O_in, LSE_in, O, LSE);
  else
    hipLaunchKernelGGL((attn_cell_fwd_kernel<PREC, false, 1024>), dim3(grid), dim3(64 * n_fast), lds, st, d, (const char*)Q,
                       (const char*)K, (const char*)Vt, (const char*)key_ws, (const char*)table_pair, O_in, LSE_in, O, LSE);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  // slow pass, in place: continues from the fast pass's state; no producer; its LDS also holds the list of slow tiles
  if (!is16(PREC) && 64 * n_rb <= 512)
    hipLaunchKernelGGL((attn_cell_fwd_kernel<PREC, true, (is16(PREC) ? 1024 : 512)>), dim3(grid), dim3(64 * n_rb), lds_slow, st,
                       d, (const char*)Q, (const char*)K, (const char*)Vt, (const char*)key_ws, (const char*)table_pair,
                       (const float*)O, (const float*)LSE, O, LSE);
  else
    hipLaunchKernelGGL((attn_cell_fwd_kernel<PREC, true, 1024>), dim3(grid), dim3(64 * n_rb), lds_slow, st, d,
                       (const char*)Q, (const char*)K, (const char*)Vt, (const char*)key_ws, (const char*)table_pair,
                       (const float*)O, (const float*)LSE, O, LSE);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_cell_fwd(const bevr_attn_desc* d, const void* Q, const void* K, const void* Vt,
                                  const void* key_ws, const float* table_pair, const float* O_in,
                                  const float* LSE_in, float* O, float* LSE, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !K || !Vt || !key_ws || !table_pair || !O || !LSE || (O_in && !LSE_in)) return BEVR_E_NULL;
  if (d->Sp > 512) return BEVR_E_SHAPE;   // one wave per 32-row block of a BEV column, at most 16 waves
  if (!bevr_aligned16(Q) || !bevr_aligned16(K) || !bevr_aligned16(Vt) || !bevr_aligned16(O) ||
      !bevr_aligned16(table_pair) || !bevr_aligned16(key_ws) || (O_in && !bevr_aligned16(O_in)))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16>(*d, Q, K, Vt, key_ws, table_pair, O_in, LSE_in, O, LSE, st);
  if (d->precision == BEVR_PREC_F16)
    return launch<BEVR_PREC_F16>(*d, Q, K, Vt, key_ws, table_pair, O_in, LSE_in, O, LSE, st);
  if (d->precision == BEVR_PREC_BF16X3)
    return launch<BEVR_PREC_BF16X3>(*d, Q, K, Vt, key_ws, table_pair, O_in, LSE_in, O, LSE, st);
  return launch<BEVR_PREC_F32>(*d, Q, K, Vt, key_ws, table_pair, O_in, LSE_in, O, LSE, st);
}
