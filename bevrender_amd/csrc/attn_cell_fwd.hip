// Attention forward over a CELL-SORTED key segment (attn_cell.h): O = softmax_n(Q^T K scale + rpe bias) V with the
// bias of a (32 keys x 32 BEV rows) tile as ONE extra MFMA.  Same arithmetic as attn_fwd.hip
// (model/SCA_deform_attn.py:331-413 of the reference), same operand layouts, same orientation (S^T[key][query],
// query on the lane, lazily rescaled online softmax, P^T fed back as the B operand of PV).
//
// Work split: workgroup = ONE BEV column j of one (problem, head); its waves are the column's 32-row blocks, so every
// wave needs the same per-(column, key) weights W: one wave builds them for the next step (its lanes = the keys, rotating
// over the waves) while all waves compute the current one, and they are handed over in LDS with the staged K / V^T tiles
// (one barrier per 64-key step).  A wave reloads its table operand only when the chunk origin changes (3 % of the
// tiles of a cell-sorted segment).
//
// Two passes (template parameter SLOW): the fast pass computes the tiles that fit one chunk and skips the others; the slow
// pass then visits ONLY the skipped tiles (per-pair gather from the table in global memory; correct for any key set),
// continuing the softmax in place from the fast pass's (O, LSE) -- and its workgroups exit at once when their column has
// no such tile, which is the normal case for a cell-sorted segment.  One kernel with both paths in its tile loop
// spilled whole accumulators.
//
// Chaining: the keys of one softmax may be split between the region kernels (scattered keys) and this one.  With
// (O_in, LSE_in) given, the online softmax starts from that state (m = LSE_in, l = 1, o = O_in) and the result is the
// softmax over both segments; with O_in == NULL it starts empty.
#include "attn_cell.h"

namespace {

constexpr float RESCALE_THR = 2.0f;   // log2 units; as attn_fwd.hip (also the slack of LSE plane 1)

template <int PREC> struct LdsC {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int K_STRIDE = 32 * EB + 16;
  static constexpr int V_STRIDE = KT * EB + 16;
  static constexpr int K_BYTES = KT * K_STRIDE;
  static constexpr int V_BYTES = 32 * V_STRIDE;
  static constexpr int KW_BYTES = KT * 16;
  static constexpr int WL = 8 * EB;                 // bytes of one lane's chunk operand
  static constexpr int W_BYTES = 2 * 64 * WL;       // two tiles per step
  static constexpr int BUF = K_BYTES + V_BYTES + KW_BYTES + W_BYTES;
  static constexpr int KCH_ROW = 32 * EB / 16;      // 16-B chunks per K row
  static constexpr int VCH_ROW = KT * EB / 16;      // 16-B chunks per V^T row of this step
  static constexpr int CH = KT * KCH_ROW;           // chunks per tile (K rows, V^T rows): 256 / 512
  static constexpr int NCH = 2 * CH + KT;           // + one KeyW record per key
  static constexpr int NST = PREC == BEVR_PREC_BF16 ? 2 : 3;   // chunks a thread carries in registers across a step
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
    dst = L::K_BYTES + (ci / L::VCH_ROW) * L::V_STRIDE + (ci % L::VCH_ROW) * 16;
  } else {
    const int ci = g - 2 * L::CH;
    src = kws + (size_t)ci * 16;
    inc = KT * 16;
    dst = L::K_BYTES + L::V_BYTES + ci * 16;
  }
}

template <int PREC, bool SLOW>
__global__ __launch_bounds__(1024) void attn_cell_fwd_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ Vt,
    const char* __restrict__ key_ws, const char* __restrict__ table_pair, const float* __restrict__ O_in,
    const float* __restrict__ LSE_in, float* __restrict__ O, float* __restrict__ LSE) {
  typedef LdsC<PREC> L;
  constexpr int EB = L::EB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 staging buffers + one Q fragment slot per wave

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
  const int i0 = wave * 32;

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
  if constexpr (SLOW) {   // anything for this column?  (uniform: scalar loads of the tile boxes)
    bool any = false;
    for (int u = 0; u < 2 * n_step; ++u) {
      const CellTile c = make_celltile(kbox[u], jrx);
      any = any || (c.live && !c.fast);
    }
    if (!any) return;
  }

  // the Q fragment lives in LDS (own lanes' data, written and read by this wave only: no barrier) and is re-read per
  // tile: 8 (bf16) / 16 (f32) registers less to carry through the loop
  char* qslot = smem + 2 * L::BUF + (wave * 64 + lane) * (32 * EB);
  {
    Frag<PREC> qf;
    qf.load(Qh + mq * 32 * EB, hi);
    if constexpr (PREC == BEVR_PREC_BF16) {
      *reinterpret_cast<u32x4*>(qslot) = __builtin_bit_cast(u32x4, qf.v[0]);
      *reinterpret_cast<u32x4*>(qslot + 16) = __builtin_bit_cast(u32x4, qf.v[1]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(qslot + 16 * k) = f32x4{qf.v[4 * k], qf.v[4 * k + 1], qf.v[4 * k + 2], qf.v[4 * k + 3]};
    }
  }
  auto load_q = [&](Frag<PREC>& f) {
    if constexpr (PREC == BEVR_PREC_BF16) {
      f.v[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qslot));
      f.v[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qslot + 16));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qslot + 16 * k);
        f.v[4 * k] = t[0]; f.v[4 * k + 1] = t[1]; f.v[4 * k + 2] = t[2]; f.v[4 * k + 3] = t[3];
      }
    }
  };

  // ---- online-softmax state --------------------------------------------------------------------------
  f32x16 o;
  float m = 0.f, l = 0.f;
  bool first = true;   // the running max is not set yet (wave-uniform)
  if (O_in) {
    const float lse_in = LSE_in[(size_t)ph * Mp + mq];
    // rows past the grid hold -inf in the incoming plane: start them empty (their results are never read)
    if (__any(lse_in > -3.0e38f)) {
      first = false;
      m = fmaxf(lse_in, -1.0e30f);
      l = hi == 0 ? 1.f : 0.f;       // the halves' denominators are added in the epilogue
      const float* orow = O_in + ((size_t)ph * Mp + mq) * 32;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(orow + 8 * g4 + 4 * hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[4 * g4 + k] = v[k];
      }
    }
  }
  if (first) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
  }

  // ---- staging: global -> registers -> LDS, one step ahead ----------------------------------------------
  u32x4 st[L::NST];
  const char* st_src[L::NST];
  int st_inc[L::NST], st_dst[L::NST];
#pragma unroll
  for (int k = 0; k < L::NST; ++k) {
    const int g = tid + k * nt;
    st_dst[k] = -1;
    st_src[k] = Kh;
    st_inc[k] = 0;
    if (g < L::NCH) chunk_map<PREC>(g, Kh, Vh, kws, d.Np, st_src[k], st_inc[k], st_dst[k]);
  }
  auto stage_load = [&](int step) {
#pragma unroll
    for (int k = 0; k < L::NST; ++k)
      if (st_dst[k] >= 0) st[k] = *reinterpret_cast<const u32x4*>(st_src[k] + (size_t)step * st_inc[k]);
  };
  auto stage_store = [&](int buf, int step) {
    char* base = smem + buf * L::BUF;
#pragma unroll
    for (int k = 0; k < L::NST; ++k)
      if (st_dst[k] >= 0) *reinterpret_cast<u32x4*>(base + st_dst[k]) = st[k];
    // small workgroups (few row blocks): the chunks beyond the registers' share are copied through directly
    for (int g = tid + L::NST * nt; g < L::NCH; g += nt) {
      const char* src;
      int inc, dst;
      chunk_map<PREC>(g, Kh, Vh, kws, d.Np, src, inc, dst);
      *reinterpret_cast<u32x4*>(base + dst) = *reinterpret_cast<const u32x4*>(src + (size_t)step * inc);
    }
  };
  // the weights of tile `t` of step `step`, built by this wave (lane & 31 = key) into buffer `buf`
  auto build_w = [&](int buf, int step, int t, const KeyW& kw) {
    const StepBox sb = kbox[2 * step + t];
    const CellTile ct = make_celltile(sb, jrx);
    float tcol, trow;
    cell_coords(kw, jrx, ct.x0, step * KT + t * 32 + lq >= d.N, tcol, trow);
    const CellFrag<PREC> w = cell_weights<PREC>(tcol, trow, hi);
    char* dst = smem + buf * L::BUF + L::K_BYTES + L::V_BYTES + L::KW_BYTES + (t * 64 + lane) * L::WL;
    if constexpr (PREC == BEVR_PREC_BF16) {
      *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, w.v);
    } else {
      *reinterpret_cast<f32x4*>(dst) = f32x4{w.v[0], w.v[1], w.v[2], w.v[3]};
      *reinterpret_cast<f32x4*>(dst + 16) = f32x4{w.v[4], w.v[5], w.v[6], w.v[7]};
    }
  };
  auto load_kw = [&](int step, int t) {
    return *reinterpret_cast<const KeyW*>(kws + ((size_t)step * KT + t * 32 + lq) * sizeof(KeyW));
  };
  // which wave builds tile t of a step: rotates, so that the extra work is spread evenly
  auto builder_of = [&](int step, int t) { return (2 * step + t) % n_wave; };

  stage_load(0);
  stage_store(0, 0);
  if constexpr (!SLOW) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (builder_of(0, t) == wave) build_w(0, 0, t, load_kw(0, t));
  }
  __syncthreads();

  // table operand of the chunk this wave holds, and its origin
  CellFrag<PREC> tf;
  int tag_x = 1 << 30, tag_a = 1 << 30;
  if constexpr (PREC == BEVR_PREC_BF16) tf.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  else {
#pragma unroll
    for (int k = 0; k < 8; ++k) tf.v[k] = 0.f;
  }

  for (int step = 0; step < n_step; ++step) {
    const int buf = step & 1;
    const char* base = smem + buf * L::BUF;
    const bool more = step + 1 < n_step;
    if (more) stage_load(step + 1);
    // the next step's key records of the tiles this wave builds (global loads, consumed after this step's tiles)
    KeyW kwn[2];
    bool bld[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bld[t] = !SLOW && more && builder_of(step + 1, t) == wave;
      kwn[t] = KeyW{0, 0.f, 0.f, 0};
      if (bld[t]) kwn[t] = load_kw(step + 1, t);
    }
    const KeyW* kwl = reinterpret_cast<const KeyW*>(base + L::K_BYTES + L::V_BYTES);

#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const StepBox sb = kbox[2 * step + t];
      const CellTile ct = make_celltile(sb, jrx);
      if (!ct.live || (bool)ct.fast == SLOW) continue;   // no unmasked key in this half / the other pass's tile (uniform)

      Frag<PREC> kf, vf;
      kf.load(base + (t * 32 + lq) * L::K_STRIDE, hi);
      load_perm(vf, base + L::K_BYTES + lq * L::V_STRIDE + t * 32 * EB, hi);
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = -m;
      {
        Frag<PREC> qf;
        load_q(qf);
        s = mma_frag(kf, qf, s);   // S^T - m
      }

      if constexpr (!SLOW) {
        if (ct.x0 != tag_x || ct.a0 != tag_a) {   // uniform
          tf = cell_table<PREC>(tbl, d, ct.x0, ct.a0 + i0 + lq, hi);
          tag_x = ct.x0;
          tag_a = ct.a0;
        }
        CellFrag<PREC> wf;
        const char* wsrc = base + L::K_BYTES + L::V_BYTES + L::KW_BYTES + (t * 64 + lane) * L::WL;
        if constexpr (PREC == BEVR_PREC_BF16) {
          wf.v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wsrc));
        } else {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wsrc), w1 = *reinterpret_cast<const f32x4*>(wsrc + 16);
          wf.v[0] = w0[0]; wf.v[1] = w0[1]; wf.v[2] = w0[2]; wf.v[3] = w0[3];
          wf.v[4] = w1[0]; wf.v[5] = w1[1]; wf.v[6] = w1[2]; wf.v[7] = w1[3];
        }
        s = mma_cell(wf, tf, s);   // + bias^T[key][query]
      } else {
        // the tile's taps do not fit one chunk: per-pair gather from the table in global memory (any key set)
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
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight (rare path, register budget)
        }
      }
      // mask padded keys (only the last step can hold any)
      if (step == n_step - 1 && d.N < d.Np) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (step * KT + t * 32 + crow(r, hi) >= d.N) ? BEVR_NEG_BIG : s[r];
      }

      // online softmax with a lazily updated running max: s holds S - m
      float tm = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tm = fmaxf(tm, s[r]);
      if (first || __any(tm > RESCALE_THR)) {   // wave-uniform: rare after the first tiles
        tm = fmaxf(tm, __shfl_xor(tm, 32));     // the lane halves hold the same queries, different keys
        const float up = first ? tm : fmaxf(tm, 0.f);
        const float al = fast_exp2(-up);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[r] *= al; s[r] -= up; }
        l *= al;
        m += up;
        first = false;
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
      o = mma_acc_b(vf, s, o);
    }

    if (more) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (bld[t]) build_w(buf ^ 1, step + 1, t, kwn[t]);
      stage_store(buf ^ 1, step + 1);
    }
    __syncthreads();
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
      // plane 1: upper bound of log2 of the row's largest softmax weight (every logit <= m + RESCALE_THR)
      Lh[(size_t)n_ph * Mp + mq] = RESCALE_THR - __log2f(lt);
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* K, const void* Vt, const void* key_ws,
           const float* table_pair, const float* O_in, const float* LSE_in, float* O, float* LSE, hipStream_t st) {
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * d.S;
  const int n_wave = d.Sp / 32;
  const size_t lds = 2 * LdsC<PREC>::BUF + (size_t)n_wave * 64 * 32 * LdsC<PREC>::EB;
  hipLaunchKernelGGL((attn_cell_fwd_kernel<PREC, false>), dim3(grid), dim3(64 * n_wave), lds, st, d, (const char*)Q,
                     (const char*)K, (const char*)Vt, (const char*)key_ws, (const char*)table_pair, O_in, LSE_in, O, LSE);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  // slow pass, in place: continues from the fast pass's state
  hipLaunchKernelGGL((attn_cell_fwd_kernel<PREC, true>), dim3(grid), dim3(64 * n_wave), lds, st, d, (const char*)Q,
                     (const char*)K, (const char*)Vt, (const char*)key_ws, (const char*)table_pair, (const float*)O,
                     (const float*)LSE, O, LSE);
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
  return launch<BEVR_PREC_F32>(*d, Q, K, Vt, key_ws, table_pair, O_in, LSE_in, O, LSE, st);
}
