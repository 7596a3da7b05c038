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
//     conflict-free LDS reads from the workgroup's table region (attn_tile.h), or two coalesced
//     256-byte global rows on the fallback path.
// Work split: workgroup = 8 waves = one 32-row block x 8 BEV columns, one column per wave; all waves walk
// the keys together, 64 per step, K / V^T / key constants staged through LDS (double buffered).
// Everything that depends only on (column, key) -- the tap column, the window offset, fx, the y weights --
// is computed once per step by each wave for its own column with lane = key (64 keys = one wave-wide
// pass) and read back as a 16-byte LDS broadcast, so the per-pair work is: one broadcast read, one tap
// read, ~5 VALU for the bias, ~5 for the softmax.  In bf16 mode the table window holds bf16 (T[y], T[y+1])
// pairs and the y-interpolation is one v_dot2c_f32_bf16 per tap column.
// blockIdx is remapped so that all query tiles of one (problem, head) run on one XCD and stream the same K/V
// through that XCD's L2.
#include "attn_tile.h"

#ifdef BEVR_PROF
__device__ unsigned long long bevr_prof_fwd[16];
extern "C" int bevr_debug_prof_fwd(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(bevr_prof_fwd), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bevr_prof_fwd), 16 * 8);
}
__device__ __forceinline__ unsigned long long prof_now_f(float dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}
#define PROF_TD(var, dep) const unsigned long long var = prof_now_f(dep)
#define PROF_ADD(i, v) pacc[i] += (v)
#else
#define PROF_TD(var, dep)
#define PROF_ADD(i, v)
#endif

namespace {

constexpr float RESCALE_THR = 2.0f;  // log2 units: lazy running-max update (P <= 2^2); also the slack of LSE plane 1 in BF16 mode
// fp16 mode: the reference sits Half::SHIFT = 10 binades under the running maximum (weights up to 2^12, small ones
// still normal numbers): threshold and plane-1 slack move up by the same amount
template <int PREC> constexpr float rescale_thr() { return PREC == BEVR_PREC_F16 ? RESCALE_THR + 10.0f : RESCALE_THR; }
template <int PREC> constexpr float ref_shift() { return PREC == BEVR_PREC_F16 ? 10.0f : 0.0f; }
constexpr int TF = 512;              // threads per workgroup
constexpr int NWF = TF / 64;         // waves = query columns per workgroup
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

template <int PREC> struct Lds {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int WIN_COLS = region_cap_fwd(PREC);         // table-region columns
  static constexpr int ENT = is16(PREC) ? 4 : 8;    // bytes per window entry: (T[y], T[y+1]) as a 16-bit pair / f32x2
  static constexpr int K_STRIDE = 32 * EB + 16;   // bytes per key row (+16: bank spread)
  static constexpr int V_STRIDE = KT * EB + 16;   // bytes per channel row
  static constexpr int K_BYTES = KT * K_STRIDE;
  static constexpr int V_BYTES = 32 * V_STRIDE;
  static constexpr int C_BYTES = KT * 16 + 32;    // KeyW per key + WinInfo
  static constexpr int BUF = K_BYTES + V_BYTES + C_BYTES;
  static constexpr int WIN = WIN_COLS * WIN_PITCH * ENT;
  static constexpr int PCK = NWF * KT * (is16(PREC) ? 16 : 32);   // per-wave (column, key) constants
  static constexpr int QL = is16(PREC) ? NWF * 128 * 16 : 16;       // 16-bit modes: Q fragments
  static constexpr int TOTAL = 2 * BUF + WIN + PCK + QL;
};

template <int PREC>
__global__ __launch_bounds__(TF, is16(PREC) ? 4 : 2) void attn_fwd_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ Vt,
    const char* __restrict__ key_ws, const char* __restrict__ table_pair,
    float* __restrict__ O, float* __restrict__ LSE BEVR_DROP_PARAMS) {
  typedef Lds<PREC> L;
  constexpr int EB = L::EB;
  constexpr int ENT = L::ENT;
  constexpr int WIN_COLS = L::WIN_COLS;
  __shared__ __attribute__((aligned(16))) char smem[2 * L::BUF];
  __shared__ __attribute__((aligned(16))) char win[L::WIN];
  typedef ColKeyT<PREC> CK;
  __shared__ __attribute__((aligned(16))) CK pck_all[NWF * KT];
  __shared__ __attribute__((aligned(16))) char qlds[L::QL];

  // ---- which (problem, head, query tile) -------------------------------------------------------
  const int n_rb = d.Sp / 32;
  const int n_cb = (d.S + NWF - 1) / NWF;
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
  CK* pck = pck_all + wave * KT;
  const int Mp = d.S * d.Sp;
  const int i0 = rb * 32;

  const char* Qh = Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB;
  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = Vt + ((size_t)ph * 32) * d.Np * EB;
  const int pg = prob * d.groups + grp;
  const KeyW* kws = reinterpret_cast<const KeyW*>(key_ws) + (size_t)pg * d.Np;
  const StepBox* kbox = reinterpret_cast<const StepBox*>(key_ws + key_ws_box_offset(d)) + (size_t)pg * (d.Np / 32);
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const int Hp8 = d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int j_first = cb * NWF;
  const int j_last = min(j_first + NWF - 1, d.S - 1);
  const float jrx_lo = (float)j_first * rx, jrx_hi = (float)j_last * rx;

  // ---- this wave's query column ------------------------------------------------------------------
  const int jcol = j_first + wave;
  const int jc = jcol < d.S ? jcol : d.S - 1;   // columns past the grid: compute on a clamped copy, never stored
  const float jrx = (float)jc * rx;
  Frag<PREC> qf;
  qf.load(Qh + ((size_t)jc * d.Sp + i0 + lq) * 32 * EB, hi);
  if constexpr (is16(PREC)) {   // 16-bit modes: the Q fragment lives in LDS (re-read per tile), not in 8 registers
    u32x4* ql = reinterpret_cast<u32x4*>(qlds) + wave * 128 + lane;
    ql[0] = __builtin_bit_cast(u32x4, qf.v[0]);
    ql[64] = __builtin_bit_cast(u32x4, qf.v[1]);
  }
  const int rowoff = (i0 + lq) * 8;
  const int lqe = lq * ENT;

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = 0.f, l = 0.f;   // running max (log2 units) and denominator; the first tile sets m
  float smax = -3.0e38f;    // F32 mode only: the largest logit seen (absolute), for LSE plane 1

  // ---- staging: global -> registers -> LDS ------------------------------------------------------
  constexpr int KCH_ROW = 32 * EB / 16;            // 16-B chunks per K row
  constexpr int VCH_ROW = KT * EB / 16;            // 16-B chunks per V^T row (this step's keys)
  constexpr int CH = KT * KCH_ROW;                 // chunks per tile: 256 (bf16) / 512 (f32)
  static_assert(CH <= TF && 32 * VCH_ROW == CH, "staging shape");
  // the 2 CH chunks of the K and V^T tiles are dealt over the TF threads (1 each in bf16 mode, 2 in f32 mode):
  // per-thread source pointer (advanced by a fixed stride per step) and LDS destination offset
  constexpr int NCHK = 2 * CH / TF;
  static_assert(NCHK * TF == 2 * CH, "staging deal");
  u32x4 st[NCHK];
  u32x4 st_kw = {0, 0, 0, 0};   // last wave: a key's constants
  const int n_step = d.Np / KT;
  // a chunk's tile (K or V^T) is uniform over its wave: uniform base pointer + 32-bit per-thread byte offset
  const char* st_base[NCHK];
  unsigned st_off[NCHK];
  int st_inc[NCHK], st_dst[NCHK];
#pragma unroll
  for (int k = 0; k < NCHK; ++k) {
    const int g = tid + k * TF;
    const int kind = __builtin_amdgcn_readfirstlane(g / CH), ci = g % CH;
    st_base[k] = kind ? Vh : Kh;
    st_inc[k] = kind ? KT * EB : CH * 16;
    if (kind == 0) {
      st_off[k] = (unsigned)ci * 16;
      st_dst[k] = (ci / KCH_ROW) * L::K_STRIDE + (ci % KCH_ROW) * 16;
    } else {
      st_off[k] = (unsigned)(((size_t)(ci / VCH_ROW) * d.Np) * EB + (ci % VCH_ROW) * 16);
      st_dst[k] = L::K_BYTES + (ci / VCH_ROW) * L::V_STRIDE + (ci % VCH_ROW) * 16;
    }
  }
  const int kt = tid - (TF - 64);   // key slot of the last wave's lanes

  auto stage_load = [&](int step) {
#pragma unroll
    for (int k = 0; k < NCHK; ++k)
      st[k] = *reinterpret_cast<const u32x4*>(st_base[k] + (size_t)step * st_inc[k] + st_off[k]);
    if (kt >= 0) {   // uniform base + laundered lane offset: no per-thread 64-bit pointer to keep alive (or spill)
      int ko = kt * 16;
      asm volatile("" : "+v"(ko));
      st_kw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(kws + (size_t)step * KT) + ko);
    }
  };
  auto stage_store = [&](int buf, int step) {
    char* base = smem + buf * L::BUF;
#pragma unroll
    for (int k = 0; k < NCHK; ++k) *reinterpret_cast<u32x4*>(base + st_dst[k]) = st[k];
    if (kt >= 0) *reinterpret_cast<u32x4*>(base + L::K_BYTES + L::V_BYTES + kt * 16) = st_kw;   // the last wave
  };

  stage_load(0);
  stage_store(0, 0);
  __syncthreads();

  Region rg;
  rg.ax0 = -(1 << 28);   // nothing contained: the first windowed step anchors
  rg.ay0 = 0;

  // boxes of the two 32-key halves of a step: uniform (scalar loads), fetched one step ahead
  StepBox sb_cur[2] = {kbox[0], kbox[1]};
  StepBox sb_nxt[2] = {kbox[2 * min(1, n_step - 1)], kbox[2 * min(1, n_step - 1) + 1]};
#ifdef BEVR_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int step = 0; step < n_step; ++step) {
    PROF_TD(t0, 0.f);
    const int buf = step & 1;
    const char* base = smem + buf * L::BUF;
    if (step + 1 < n_step) stage_load(step + 1);
    const KeyW* kws = reinterpret_cast<const KeyW*>(base + L::K_BYTES + L::V_BYTES);
    const StepBox sb0 = sb_cur[0], sb1 = sb_cur[1];
    sb_cur[0] = sb_nxt[0];
    sb_cur[1] = sb_nxt[1];
    sb_nxt[0] = kbox[2 * min(step + 2, n_step - 1)];
    sb_nxt[1] = kbox[2 * min(step + 2, n_step - 1) + 1];
    // if the whole step's box fits (the common case) both halves share one region test and one pass of per-key
    // constants; otherwise the halves are windowed separately
    const WinInfo wi_step = make_wininfo(box_union(sb0, sb1), jrx_lo, jrx_hi, WIN_COLS);
    auto move_region = [&](const WinInfo& wi, bool mid_step) {
      BEVR_ASSERT_WG_UNIFORM(wi.xlo * 131 + wi.amin * 7 + wi.ncols + (int)mid_step);
      if (mid_step) __syncthreads();   // every wave must be done with the first half's taps
      rg = region_anchor(wi, d, i0, WIN_COLS);
      // fill the region: one wave-wide load per table column (lane = row)
      const size_t y0 = (size_t)(i0 + rg.ay0 + d.y_off) + lane;
      BEVR_ASSERT(i0 + rg.ay0 + d.y_off >= 0 && i0 + rg.ay0 + d.y_off + WIN_PITCH <= d.Hp && rg.ax0 + d.x_off >= 0);
      // all of a wave's loads are issued before the first LDS store: one L2 latency per move, not one per column
      constexpr int PER_WAVE = WIN_COLS / NWF;
      static_assert(PER_WAVE * NWF == WIN_COLS, "region columns are dealt evenly over the waves");
      f32x2 fv[PER_WAVE];
#pragma unroll
      for (int k = 0; k < PER_WAVE; ++k) fv[k] = region_entry(tbl, d, rg, wave + k * NWF, y0);
#pragma unroll
      for (int k = 0; k < PER_WAVE; ++k) {
        const int c = wave + k * NWF;
        if constexpr (is16(PREC))
          *reinterpret_cast<unsigned*>(win + (c * WIN_PITCH + lane) * ENT) = Half<PREC>::pack2(fv[k][0], fv[k][1]);
        else
          *reinterpret_cast<f32x2*>(win + (c * WIN_PITCH + lane) * ENT) = fv[k];
      }
      __syncthreads();
    };
    // this wave's (column, key) constants: key slot `ki` of the step, whose half starts at table row `amin_h`
    auto key_consts = [&](int ki, int amin_h) {
      const KeyW kw = kws[ki];
      const float tx = jrx + (kw.b - (float)rg.ax0);
      const float xf = floorf(tx);
      CK e;
      const float fx = tx - xf, fy = kw.fy;
      e.set((1.0f - fx) * (1.0f - fy), (1.0f - fx) * fy, fx * (1.0f - fy), fx * fy);
      e.cell = (int)xf * (WIN_PITCH * ENT) + ((kw.arow8 >> 3) + (amin_h - rg.ay0)) * ENT;   // arow8: relative to the half
      pck[ki] = e;   // read back by this wave only: LDS operations of a wave execute in order
    };
    if (wi_step.ok) {
      if (!region_contains(rg, wi_step, WIN_COLS)) move_region(wi_step, false);
      key_consts(lane, hi ? sb1.amin : sb0.amin);   // lane = key
    }
    PROF_TD(t1, 0.f);
    PROF_ADD(0, t1 - t0);

#pragma unroll
    for (int ks = 0; ks < KT / 32; ++ks) {
      bool use_win = wi_step.ok != 0;   // workgroup-uniform
      if (!wi_step.ok) {
        const StepBox sbh = ks ? sb1 : sb0;
        const WinInfo wi = make_wininfo(sbh, jrx_lo, jrx_hi, WIN_COLS);
        use_win = wi.ok != 0;
        if (use_win) {
          if (!region_contains(rg, wi, WIN_COLS)) move_region(wi, ks != 0);
          key_consts(ks * 32 + lq, sbh.amin);   // lane & 31 = key (both lane halves write the same value)
        }
      }
      PROF_TD(t2, 0.f);
      Frag<PREC> kf, vf;
      kf.load(base + (ks * 32 + lq) * L::K_STRIDE, hi);
      load_perm(vf, base + L::K_BYTES + lq * L::V_STRIDE + ks * 32 * EB, hi);

      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = -m;
      if constexpr (is16(PREC)) {   // own lanes' data, written by this wave: no barrier needed
        const u32x4* ql = reinterpret_cast<const u32x4*>(qlds) + wave * 128 + lane;
        Frag<PREC> qs;
        qs.v[0] = __builtin_bit_cast(bf16x8, ql[0]);
        qs.v[1] = __builtin_bit_cast(bf16x8, ql[64]);
        s = mma_frag(kf, qs, s);   // S^T - m
      } else {
        s = mma_frag(kf, qf, s);
      }

      PROF_TD(t3, s[0] + s[15]);
      PROF_ADD(1, t3 - t2);
      // relative-position bias: rows of the tile are keys crow(r, hi); lanes are BEV rows i0 + lq.
      if (use_win) {
        const char* wl = win + lqe;
        if constexpr (is16(PREC)) {
          // hand-pipelined, BIAS_BATCH key rows at a time: all cell reads, then all tap and weight reads, then the dot
          // products -- two LDS latencies per batch instead of two per two or three rows (the compiler's own schedule)
#ifndef BEVR_BIAS_BATCH
#define BEVR_BIAS_BATCH 8
#endif
          constexpr int BIAS_BATCH = BEVR_BIAS_BATCH;
#pragma unroll
          for (int r0 = 0; r0 < 16; r0 += BIAS_BATCH) {
            // scalars, not 2-vectors: ROCm 7.2 feeds element 0 of a 2-vector to both dot products
            int cell[BIAS_BATCH];
            unsigned wa[BIAS_BATCH], wb[BIAS_BATCH], ta[BIAS_BATCH], tb[BIAS_BATCH];
#pragma unroll
            for (int k = 0; k < BIAS_BATCH; ++k) {
              const CK& e = pck[ks * 32 + crow(r0 + k, hi)];
              cell[k] = e.cell;
              wa[k] = e.wA;
              wb[k] = e.wB;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < BIAS_BATCH; ++k) {
              const char* p = wl + cell[k];
              ta[k] = *reinterpret_cast<const unsigned*>(p);
              tb[k] = *reinterpret_cast<const unsigned*>(p + WIN_PITCH * ENT);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < BIAS_BATCH; ++k) {
              const float sv = Half<PREC>::dot2(ta[k], wa[k], s[r0 + k]);
              s[r0 + k] = Half<PREC>::dot2(tb[k], wb[k], sv);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const CK e = pck[ks * 32 + crow(r, hi)];
            const char* p = wl + e.cell;
            const f32x2 t0 = *reinterpret_cast<const f32x2*>(p);
            const f32x2 t1 = *reinterpret_cast<const f32x2*>(p + WIN_PITCH * ENT);
            s[r] = fmaf(t1[1], e.w11(), fmaf(t1[0], e.w10(), fmaf(t0[1], e.w01(), fmaf(t0[0], e.w00(), s[r]))));
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const KeyW c = kws[ks * 32 + crow(r, hi)];
          const float wy0 = 1.0f - c.fy;
          float tx = jrx + c.b;
          float xf = floorf(tx);
          float fx = tx - xf;
          unsigned off = (unsigned)((int)xf * Hp8 + c.aoff + rowoff);
          f32x2 t0 = *reinterpret_cast<const f32x2*>(tbl + off);
          f32x2 t1 = *reinterpret_cast<const f32x2*>(tbl + off + Hp8);
          float u0 = t0[0] * wy0 + t0[1] * c.fy;
          float u1 = t1[0] * wy0 + t1[1] * c.fy;
          s[r] += u0 + fx * (u1 - u0);
        }
      }
      PROF_TD(t4, s[0] + s[15] + s[7]);
      PROF_ADD(2, t4 - t3);
      // mask padded keys (only the last step can hold any)
      if (step == n_step - 1 && d.N < d.Np) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (step * KT + ks * 32 + crow(r, hi) >= d.N) ? BEVR_NEG_BIG : s[r];
      }

      // online softmax with a lazily updated running max: s holds S - m
      float tm = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tm = fmaxf(tm, s[r]);
      if constexpr (!is16(PREC)) smax = fmaxf(smax, tm + m);
      const bool first = (step == 0 && ks == 0);
      if (first || __any(tm > rescale_thr<PREC>())) {   // wave-uniform: rare after the first tiles
        // the two lane halves hold the same queries (different keys): agree on the maximum only when it is needed
        tm = fmaxf(tm, __shfl_xor(tm, 32)) - ref_shift<PREC>();
        const float up = first ? tm : fmaxf(tm, 0.f);   // the max only moves up, except when it is first set
        // first tile: o and l are still zero and exp2(-up) may be +inf (all logits below -127: 0 * inf would be NaN;
        // found by the config-5 run, where the history BEV grows over 6 frames)
        const float al = first ? 0.f : fast_exp2(-up);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[r] *= al; s[r] -= up; }
        l *= al;
        m += up;
      }
      f32x2 ls2 = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 pp = {fast_exp2(s[r]), fast_exp2(s[r + 1])};
        s[r] = pp[0];
        s[r + 1] = pp[1];
        ls2 += pp;
      }
      l += ls2[0] + ls2[1];
#if BEVR_DROP
      {
        // dropout acts on the normalised weights: the denominator above is of the unmasked ones; a kept weight is scaled
        const uint32_t hrow = bevr_drop_row(drop_seed, (uint32_t)ph, (uint32_t)(jc * d.Sp + i0 + lq));
        const float ksc = 65536.0f / (65536.0f - (float)drop_thr);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s[r] = bevr_drop_keep(hrow, (uint32_t)(step * KT + ks * 32 + crow(r, hi)), drop_thr) ? s[r] * ksc : 0.f;
      }
#endif
      PROF_TD(t5, l + s[3]);
      PROF_ADD(3, t5 - t4);
      o = mma_acc_b(vf, s, o);
      PROF_TD(t6, o[0] + o[15]);
      PROF_ADD(4, t6 - t5);
    }
    PROF_TD(t7, 0.f);

    if (step + 1 < n_step) stage_store(buf ^ 1, step + 1);
    __syncthreads();
    PROF_TD(t8, 0.f);
    PROF_ADD(5, t8 - t7);
    PROF_ADD(6, t8 - t0);
    PROF_ADD(7, 1);
  }
#ifdef BEVR_PROF
  if (lane == 0 && (wave == 0 || wave == NWF - 1)) {
    for (int i = 0; i < 8; ++i) atomicAdd(&bevr_prof_fwd[(wave ? 8 : 0) + i], pacc[i]);
  }
#endif

  // ---- epilogue: normalise, store O^T tile as [q][32] rows and the log2-sum-exp ------------------
  if (jcol < d.S) {
    float* Oh = O + ((size_t)ph * Mp) * 32;
    float* Lh = LSE + (size_t)ph * Mp;
    float lt = l + __shfl_xor(l, 32);
    float inv = 1.0f / lt;
    size_t mq = (size_t)jcol * d.Sp + i0 + lq;
    float* orow = Oh + mq * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = o[4 * g4 + k] * inv;
      *reinterpret_cast<f32x4*>(orow + 8 * g4 + 4 * hi) = v;
    }
    const float lse = m + __log2f(lt);
    const float smax_row = fmaxf(smax, __shfl_xor(smax, 32));   // the lane halves saw different keys
    if (hi == 0) {
      Lh[mq] = lse;
      // plane 1: an upper bound of log2 of the row's largest softmax weight, for the backward's fixed-point scale.
      // BF16 mode: every logit of the row is <= m + RESCALE_THR (a tile above that moves m), so P <= 2^THR / l -- free,
      // where tracking the exact maximum costs a register and 7 % of the kernel at 128 VGPRs.  F32 mode (the parity
      // mode, 212 VGPRs): the exact maximum.
      Lh[(size_t)n_ph * Mp + mq] = !is16(PREC) ? smax_row - lse : rescale_thr<PREC>() - __log2f(lt);
    }
  }
}

template <int PREC>
int launch_fwd(const bevr_attn_desc& d, const void* Q, const void* K, const void* Vt, const void* key_ws,
               const float* table_pair, float* O, float* LSE, hipStream_t st BEVR_DROP_PARAMS) {
  const int n_rb = d.Sp / 32, n_cb = (d.S + NWF - 1) / NWF;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_rb * n_cb;
  hipLaunchKernelGGL((attn_fwd_kernel<PREC>), dim3(grid), dim3(TF), 0, st, d, (const char*)Q, (const char*)K,
                     (const char*)Vt, (const char*)key_ws, (const char*)table_pair, O, LSE BEVR_DROP_ARGS);
  return (int)hipGetLastError();
}

}  // namespace

#if BEVR_DROP
extern "C" int bevr_attn_fwd_dropout(const bevr_attn_desc* d, const void* Q, const void* K, const void* Vt,
                                     const void* key_ws, const float* table_pair, float* O,
                                     float* LSE, unsigned drop_thr, unsigned drop_seed, void* stream) {
  if (drop_thr >= 65536u) return BEVR_E_SHAPE;
#else
extern "C" int bevr_attn_fwd(const bevr_attn_desc* d, const void* Q, const void* K, const void* Vt,
                             const void* key_ws, const float* table_pair, float* O,
                             float* LSE, void* stream) {
#endif
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !K || !Vt || !key_ws || !table_pair || !O || !LSE) return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(K) || !bevr_aligned16(Vt) || !bevr_aligned16(O) ||
      !bevr_aligned16(table_pair))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16) return launch_fwd<BEVR_PREC_BF16>(*d, Q, K, Vt, key_ws, table_pair, O, LSE, st BEVR_DROP_ARGS);
  if (d->precision == BEVR_PREC_F16) return launch_fwd<BEVR_PREC_F16>(*d, Q, K, Vt, key_ws, table_pair, O, LSE, st BEVR_DROP_ARGS);
  if (d->precision == BEVR_PREC_BF16X3)
    return launch_fwd<BEVR_PREC_BF16X3>(*d, Q, K, Vt, key_ws, table_pair, O, LSE, st BEVR_DROP_ARGS);
  return launch_fwd<BEVR_PREC_F32>(*d, Q, K, Vt, key_ws, table_pair, O, LSE, st BEVR_DROP_ARGS);
}
