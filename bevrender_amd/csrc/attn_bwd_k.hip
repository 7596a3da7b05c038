// Attention backward, key side: dK, dV and the gradients of the keys' table coordinates (a_n, b_n).
// Key-stationary: a workgroup owns 256 consecutive keys and sweeps all query tiles, so every per-key sum
// stays in registers and nothing is reduced across workgroups.  Orientation here is "key on the lane":
// tiles are S[query][key], which makes P and dS the B operands of
//   dV^T[c][key] += dO^T[c][q] P[q][key]      dK^T[c][key] += Q^T[c][q] dS[q][key]
// without any transpose, and makes the key's table coordinates lane constants: the 16 registers of a
// tile are 16 consecutive-ish BEV rows of one column, so the bilinear taps are reached with immediate
// offsets from one per-lane base address.
// Row constants (-LSE[q], -delta[q]) are preloaded into the accumulators of S and dP.
//
// Two kernels cover the key blocks between them (each exits at once on the other's blocks):
//  * attn_bwd_k_win_kernel: 8 waves x 32 keys, 4 waves per SIMD.  The sweep runs column-major over the BEV
//    grid, so for one BEV column j the block's 256 keys touch the table only in
//      rows    [Amin, Amax + Sp]                    (all 32-row tiles of the column)
//      columns [floor(j rx + bmin), floor(j rx + bmax) + 1]
//    That slab is kept in LDS as a ring of table columns (f32, one value per entry): moving to column j + 1
//    shifts the range right by rx, so only ~rx new table columns are fetched per BEV column.  Taps are then
//    LDS reads at immediate offsets from two per-lane bases.
//  * attn_bwd_k_gather_kernel: the general path (4 waves x 64 keys, taps gathered from L2) for blocks whose
//    slab does not fit the ring (keys of the block far apart in table space).
#include "attn_kstage.h"

#ifdef BEVR_PROF
__device__ unsigned long long bevr_prof_k[16];
extern "C" int bevr_debug_prof_k(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(bevr_prof_k), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bevr_prof_k), 16 * 8);
}
__device__ __forceinline__ unsigned long long prof_now(float dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}
#define PROF_TD(var, dep) const unsigned long long var = prof_now(dep)
#define PROF_ADD(i, v) pacc[i] += (v)
#else
#define PROF_TD(var, dep)
#define PROF_ADD(i, v)
#endif

namespace {

constexpr int KEYS_WG = 384;    // keys per workgroup, both kernels: the unit of ownership

// bounding box of a key block in table coordinates (padded keys excluded)
struct KBox { int amin, amax; float bmin, bmax; };

__device__ __forceinline__ int wred_min_i(int v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = min(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ int wred_max_i(int v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = max(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float wred_min_f(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = fminf(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float wred_max_f(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}

// Workgroup-wide box from per-lane partial boxes (empty: lo > hi).  red: n_wave KBox slots in LDS.  Ends with a
// barrier; the result is uniform.
__device__ __forceinline__ KBox wg_key_box(int a_lo, int a_hi, float b_lo, float b_hi, KBox* red, int n_wave, int wave,
                                           int lane) {
  KBox w;
  w.amin = wred_min_i(a_lo);
  w.amax = wred_max_i(a_hi);
  w.bmin = wred_min_f(b_lo);
  w.bmax = wred_max_f(b_hi);
  if (lane == 0) red[wave] = w;
  __syncthreads();
  KBox r = red[0];
  for (int k = 1; k < n_wave; ++k) {
    const KBox o = red[k];
    r.amin = min(r.amin, o.amin);
    r.amax = max(r.amax, o.amax);
    r.bmin = fminf(r.bmin, o.bmin);
    r.bmax = fmaxf(r.bmax, o.bmax);
  }
  return r;
}

// Ring geometry of a block's table slab; `fits` decides which kernel owns the block.
struct Slab {
  int rows;     // Amax - Amin + Sp + 1: window row w is padded table row Amin + y_off + w
  int pitch;    // rows rounded up to 7 (mod 32): spreads the lanes' columns over the banks
  int ncw;      // ring capacity in table columns
  bool fits;
};
__device__ __forceinline__ Slab make_slab(const KBox& kb, const bevr_attn_desc& d, int wcap, float rx_ceil) {
  Slab sl;
  if (kb.amax < kb.amin) {   // no live key in the block: nothing to compute; the window kernel owns it
    sl.rows = 1; sl.pitch = 1; sl.ncw = 1; sl.fits = true;
    return sl;
  }
  sl.rows = kb.amax - kb.amin + d.Sp + 1;
  // pitch = 7 (mod 32): a compact cluster of keys (column slot s, row a) then lands on banks 7 s + a, nearly all
  // distinct -- measured 4 % faster than an arbitrary odd pitch (bank conflicts were 36 % of the LDS-active cycles)
  sl.pitch = sl.rows + ((7 - sl.rows) & 31);
  sl.ncw = wcap / sl.pitch - 1;   // the last column of the LDS array is the "kill" column of padded keys
  // columns of one BEV column's range plus those of the next (prefetched while this one is in use), plus slack
  const int need = (int)floorf(kb.bmax - kb.bmin) + 4 + (int)rx_ceil + 1;
  sl.fits = sl.ncw >= need;
  return sl;
}

// clamp a key's table coordinates exactly as make_keyc does and split off the integer row
__device__ __forceinline__ void key_split(float a, float b, const bevr_attn_desc& d, int& A, float& fy, float& bc) {
  const float aL = -(float)(d.Sp + 1), aU = (float)(d.Ht + 1);
  const float half = (float)((d.Wt) / 2);
  const float bL = -(half + 2.0f), bU = (float)(d.Wt + 1);
  a = fminf(fmaxf(a, aL), aU);
  bc = fminf(fmaxf(b, bL), bU);
  const float af = floorf(a);
  A = (int)af;
  fy = a - af;
}

// =========================================================================================================
// Window kernel
// =========================================================================================================
constexpr int TW = 768;   // 12 waves x 32 keys, 3 waves per SIMD (170 registers): one workgroup per CU
constexpr int KWW = 1;

template <int PREC>
__global__ __launch_bounds__(TW, 3) void attn_bwd_k_win_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ Qt, const char* __restrict__ K,
    const char* __restrict__ V, const float* __restrict__ key_a, const float* __restrict__ key_b,
    const char* __restrict__ table_pair, const char* __restrict__ dO, const char* __restrict__ dOt,
    const float* __restrict__ LSE, const float* __restrict__ delta, const float* __restrict__ grad_scale,
    float* __restrict__ dK, float* __restrict__ dV, float* __restrict__ dkey_a, float* __restrict__ dkey_b BEVR_DROP_PARAMS) {
  // NT 32-query tiles are staged and processed per barrier.  With one tile per barrier the three waves of a SIMD
  // ran in lockstep -- MFMAs together, then the VALU-bound bias / exp / gradient loop together, then the barrier --
  // and each pipe idled while the other worked; with two, the waves drift apart inside an iteration and one wave's
  // MFMAs overlap another's VALU phase, and there are half as many barriers.  (f32 mode keeps one: its tiles are
  // twice the bytes and the LDS ring leaves no room.)
  constexpr int NT = is16(PREC) ? 2 : 1;
  // fp16 mode (include/bevrender_hip.h, grad_scale[2..5]): P' = P 2^kp, dS16 = P' (dP - delta) c2
  const float kp16 = PREC == BEVR_PREC_F16 ? grad_scale[2] : 0.f, c2_16 = PREC == BEVR_PREC_F16 ? grad_scale[3] : 1.f;
  const float ds_inv = PREC == BEVR_PREC_F16 ? grad_scale[4] : 1.f, p_inv = PREC == BEVR_PREC_F16 ? grad_scale[5] : 1.f;
  typedef LdsK<PREC, NT> L;
  constexpr int EB = L::EB;
  static_assert(2 * L::BUF + L::WCAP * 4 + 256 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[2 * L::BUF];
  __shared__ __attribute__((aligned(16))) float win[L::WCAP];
  __shared__ __attribute__((aligned(16))) KBox red[TW / 64];

  const int n_kb = (d.Np + KEYS_WG - 1) / KEYS_WG;
  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / n_kb) * 8 + xcd;
  if (ph >= n_ph) return;
  const int kblk = slot % n_kb;
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int grp = hd / (d.heads / d.groups);
  const int qb = prob / d.q_div;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const int Mp = d.S * d.Sp;
  const int n_rb = d.Sp / 32;

  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = V + ((size_t)ph * d.Np) * 32 * EB;
  const float* ka = key_a + (size_t)(prob * d.groups + grp) * d.Np;
  const float* kb = key_b + (size_t)(prob * d.groups + grp) * d.Np;
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));

  // ---- this wave's KWW x 32 keys ---------------------------------------------------------------------
  bool wave_live[KWW], dead[KWW];
  int key[KWW], A[KWW];
  float fy[KWW], bcl[KWW];
  int a_lo = 0x7fffffff, a_hi = (int)0x80000000;
  float b_lo = 3.0e38f, b_hi = -3.0e38f;
#pragma unroll
  for (int w = 0; w < KWW; ++w) {
    const int k0 = kblk * KEYS_WG + (wave * KWW + w) * 32;
    wave_live[w] = k0 < d.Np;                       // wave-uniform
    key[w] = wave_live[w] ? k0 + lq : lq;           // dead sub-tiles read a valid address, never store
    dead[w] = !wave_live[w] || key[w] >= d.N;       // padded key: P = 0
    key_split(ka[key[w]], kb[key[w]], d, A[w], fy[w], bcl[w]);
    if (!dead[w]) {
      a_lo = min(a_lo, A[w]); a_hi = max(a_hi, A[w]);
      b_lo = fminf(b_lo, bcl[w]); b_hi = fmaxf(b_hi, bcl[w]);
    }
  }
  const KBox box = wg_key_box(a_lo, a_hi, b_lo, b_hi, red, TW / 64, wave, lane);
  const Slab sl = make_slab(box, d, L::WCAP, ceilf(rx));
  if (!sl.fits) {
    // the gather kernel owns this block; its workgroups split the query sweep and ADD their partial dK / dV, so
    // the rows are cleared here (this kernel runs first on the stream)
#pragma unroll
    for (int w = 0; w < KWW; ++w) {
      if (!wave_live[w]) continue;
      float* kr = dK + ((size_t)ph * d.Np + key[w]) * 32;
      float* vr = dV + ((size_t)ph * d.Np + key[w]) * 32;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *reinterpret_cast<f32x4*>(kr + 8 * g4 + 4 * hi) = z;
        *reinterpret_cast<f32x4*>(vr + 8 * g4 + 4 * hi) = z;
      }
    }
    return;
  }
  if (box.amax < box.amin) {             // only padded keys: their gradients are zero
#pragma unroll
    for (int w = 0; w < KWW; ++w) {
      if (!wave_live[w]) continue;
      float* kr = dK + ((size_t)ph * d.Np + key[w]) * 32;
      float* vr = dV + ((size_t)ph * d.Np + key[w]) * 32;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *reinterpret_cast<f32x4*>(kr + 8 * g4 + 4 * hi) = z;
        *reinterpret_cast<f32x4*>(vr + 8 * g4 + 4 * hi) = z;
      }
    }
    return;
  }
  int arel4[KWW];
  Frag<PREC> kf[KWW], vf[KWW];
  f32x16 dk[KWW], dv[KWW];
  float da[KWW], db[KWW], fx[KWW];
  int base0[KWW], base1[KWW];   // byte offsets of the lane's taps (row i = 0) in ring columns X and X + 1
#pragma unroll
  for (int w = 0; w < KWW; ++w) {
    if (dead[w]) { A[w] = box.amin; bcl[w] = box.bmin; fy[w] = 0.f; }   // padded key: taps in the kill column => P = 0
    arel4[w] = (A[w] - box.amin) * 4;                   // byte offset of the key's row 0 inside a ring column
    kf[w].load(Kh + (size_t)key[w] * 32 * EB, hi);
    vf[w].load(Vh + (size_t)key[w] * 32 * EB, hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[w][r] = 0.f; dv[w][r] = 0.f; }
    da[w] = 0.f; db[w] = 0.f; fx[w] = 0.f; base0[w] = 0; base1[w] = 0;
  }

  QStage<PREC, TW, NT> qs;
  qs.init(tid, Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB, dO + ((size_t)ph * Mp) * 32 * EB,
          Qt + ((size_t)(qb * d.heads + hd) * 32) * Mp * EB, dOt + ((size_t)ph * 32) * Mp * EB,
          LSE + (size_t)ph * Mp, delta + (size_t)ph * Mp, Mp, kp16);
  // tile t = (BEV column j = t / n_rb, row block rb = t % n_rb) is the 32 packed queries [32 t, 32 t + 32): consecutive
  // tiles are contiguous in memory.  Iteration `it` stages the NT tiles from first_tile(it); when the tile count is odd
  // the last iteration re-stages the previous tile in front of the last one (never reads past the array) and skips it.
  const int n_tile = d.S * n_rb;
  const int n_it = (n_tile + NT - 1) / NT;
  auto first_tile = [&](int it) { return NT == 1 ? it : max(0, min(NT * it, n_tile - NT)); };
  qs.load(tid, (size_t)first_tile(0) * 32);
  qs.store(tid, smem);

  // ring state (uniform): table columns [wlo, whi] are resident; column X lives in slot (X - xbase) % ncw
  const int xbase = (int)floorf(box.bmin) - 8;
  const float inv_ncw = 1.0f / (float)sl.ncw;
  auto slot_of = [&](int X) {   // X >= xbase always
    const int xs = X - xbase;
    int q = (int)((float)xs * inv_ncw);
    int r = xs - q * sl.ncw;
    r = r < 0 ? r + sl.ncw : r;
    r = r >= sl.ncw ? r - sl.ncw : r;
    return r;
  };
  int whi = -(1 << 30);
  constexpr int PF = 4;   // prefetch registers per lane: up to PF * 12 (column, 64-row chunk) units per BEV column
  const int kill_off = sl.ncw * sl.pitch * 4;   // byte offset of the kill column: -1e30 everywhere
  for (int r = tid; r < sl.pitch; r += TW) win[sl.ncw * sl.pitch + r] = BEVR_NEG_BIG;
  const int n_chunk = (sl.rows + 63) / 64;
  const size_t trow0 = (size_t)(box.amin + d.y_off);
  __syncthreads();

#ifdef BEVR_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int it = 0; it < n_it; ++it) {
    PROF_TD(t0, 0.f);
    const int buf = it & 1;
    const char* base = smem + buf * L::BUF;
    const int t_first = first_tile(it);
    if (it + 1 < n_it) qs.load(tid, (size_t)first_tile(it + 1) * 32);
    int pf_first = 0, pf_units = 0;
    float pf_v[PF];
#pragma unroll
    for (int sub = 0; sub < NT; ++sub) {
    const int t = t_first + sub;
    if (t < NT * it) continue;   // the re-staged tile of an odd tail: already processed (uniform)
    const int j = t / n_rb, rb = t - j * n_rb;
    if (rb == 0) {
      // ---- new BEV column: make its table columns resident and refresh the lanes' tap bases -----------
      const float jr = (float)j * rx;
      const int xhi = (int)floorf(jr + box.bmax) + 1;
      if (whi < xhi) {   // not prefetched (first column, or a shift too large for the prefetch registers)
        const int xlo = (int)floorf(jr + box.bmin);
        const int xfirst = max(whi + 1, xlo);
        const int n_new = xhi - xfirst + 1;
        for (int u = wave; u < n_new * n_chunk; u += TW / 64) {   // uniform trip count per wave
          const int c = u / n_chunk, ch = u - c * n_chunk;
          const int X = xfirst + c;
          const int row = ch * 64 + lane;
          if (row < sl.rows) {
            const float v = *reinterpret_cast<const float*>(tbl + ((size_t)(X + d.x_off) * d.Hp + trow0 + row) * 8);
            win[slot_of(X) * sl.pitch + row] = v;
          }
        }
        whi = xhi;
        __syncthreads();
      }
#pragma unroll
      for (int w = 0; w < KWW; ++w) {
        const float tx = jr + bcl[w];
        const float xf = floorf(tx);
        fx[w] = dead[w] ? 0.f : tx - xf;
        const int X = (int)xf;
        base0[w] = dead[w] ? kill_off : slot_of(X) * sl.pitch * 4 + arel4[w];
        base1[w] = dead[w] ? kill_off : slot_of(X + 1) * sl.pitch * 4 + arel4[w];
      }
    }
    // prefetch of the next BEV column's new table columns: loads now, LDS stores before this iteration's barrier
    // (their slots are outside the current column's range: make_slab reserved the room).  Only when the column's last
    // tile is also the iteration's last: otherwise the next column starts before that barrier and fetches its
    // columns itself (the `whi < xhi` path above, with its own barrier).
    if (sub == NT - 1 && rb == n_rb - 1 && j + 1 < d.S) {
      const float jr = (float)(j + 1) * rx;
      const int xhi = (int)floorf(jr + box.bmax) + 1;
      const int xfirst = max(whi + 1, (int)floorf(jr + box.bmin));
      const int units = (xhi - xfirst + 1) * n_chunk;
      if (units <= PF * (TW / 64)) {
        pf_first = xfirst;
        pf_units = units;
        whi = xhi;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          const int u = wave + k * (TW / 64);
          const int c = u / n_chunk, ch = u - c * n_chunk;
          const int row = ch * 64 + lane;
          pf_v[k] = 0.f;
          if (u < units && row < sl.rows)
            pf_v[k] = *reinterpret_cast<const float*>(tbl + ((size_t)(xfirst + c + d.x_off) * d.Hp + trow0 + row) * 8);
        }
      }
    }
    PROF_TD(t1, 0.f);
    PROF_ADD(0, t1 - t0);

#pragma unroll
    for (int w = 0; w < KWW; ++w) {
      if (!wave_live[w]) continue;
      const f32x4* rc = reinterpret_cast<const f32x4*>(base + 2 * L::TILE_Q + 2 * L::TILE_T) + sub * 8;
      f32x16 s, dp;
      // one operand fragment alive at a time (the register budget is 128): S first, then dP
      {
        Frag<PREC> qf;
        qf.load(base + (sub * 32 + lq) * L::STRIDE, hi);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {   // rows 8 g4 + 4 hi + 0..3
          const f32x4 l4 = rc[2 * g4 + hi];
#pragma unroll
          for (int k = 0; k < 4; ++k) s[4 * g4 + k] = l4[k];
        }
        s = mma_frag(qf, kf[w], s);      // S[q][key] - LSE[q]
      }
      {
        Frag<PREC> dof;
        dof.load(base + L::TILE_Q + (sub * 32 + lq) * L::STRIDE, hi);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 d4 = rc[NT * 8 + 2 * g4 + hi];
#pragma unroll
          for (int k = 0; k < 4; ++k) dp[4 * g4 + k] = d4[k];
        }
        dp = mma_frag(dof, vf[w], dp);   // dP[q][key] - delta[q]
      }
#if BEVR_DROP
      // dropout: dS = P (D dP - delta), dV from D P  (D = keep / (1 - p), the forward's mask: bevr_drop_keep)
      unsigned kmask = 0u;
      const float ksc = 65536.0f / (65536.0f - (float)drop_thr);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t hrow = bevr_drop_row(drop_seed, (uint32_t)ph, (uint32_t)(j * d.Sp + rb * 32 + crow(r, hi)));
        const bool keep = bevr_drop_keep(hrow, (uint32_t)key[w], drop_thr);
        const float nd = rc[NT * 8 + 2 * (r >> 2) + hi][r & 3];
        dp[r] = keep ? fmaf(ksc, dp[r] - nd, nd) : nd;
        kmask |= (keep ? 1u : 0u) << r;
      }
#endif
      PROF_TD(t2, s[0] + dp[15]);
      PROF_ADD(1, t2 - t1);

      const int ioff = (rb * 32 + 4 * hi) * 4;
      const char* p0 = reinterpret_cast<const char*>(win) + base0[w] + ioff;
      const char* p1 = reinterpret_cast<const char*>(win) + base1[w] + ioff;
      // two query rows per instruction (v_pk_*_f32): rows (k, k + 1) of a group are registers (r, r + 1), and
      // the taps are read as aligned register pairs (rows k, k+1) and (rows k+1, k+2)
      const f32x2 fy2 = {fy[w], fy[w]}, fx2 = {fx[w], fx[w]};
      f32x2 sa2 = {0.f, 0.f}, sb2 = {0.f, 0.f};
      const int n_live = d.S - rb * 32;   // query rows of this tile inside the grid (the last tile of a column is partial)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (8 * g4 >= n_live) {   // rows 8 g4 .. 8 g4 + 7 are all padding: no bias, no exp; they contribute nothing
#pragma unroll
          for (int k = 0; k < 4; ++k) { s[4 * g4 + k] = 0.f; dp[4 * g4 + k] = 0.f; }
          continue;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = 4 * g4 + 2 * h, o = (8 * g4 + 2 * h) * 4;
          const float* a0 = reinterpret_cast<const float*>(p0 + o);
          const float* a1 = reinterpret_cast<const float*>(p1 + o);
          const f32x2 ta = {a0[0], a0[1]}, tb = {a0[1], a0[2]};   // column X:     rows (k, k+1), (k+1, k+2)
          const f32x2 qa = {a1[0], a1[1]}, qb = {a1[1], a1[2]};   // column X + 1
          const f32x2 d0 = tb - ta, d1 = qb - qa;                 // d / dy of the two columns
          const f32x2 u0 = ta + fy2 * d0;                         // (1 - fy) ta + fy tb: the differences serve twice
          const f32x2 u1 = qa + fy2 * d1;
          const f32x2 du = u1 - u0;
          const f32x2 s2 = {s[r], s[r + 1]}, dp2 = {dp[r], dp[r + 1]};
          const f32x2 sv = (s2 + u0) + fx2 * du;
          const f32x2 pp = {fast_exp2(sv[0]), fast_exp2(sv[1])};   // padded keys: -1e30 from the kill column => 0
          f32x2 ds = pp * dp2;                                     // ln2 folded into the epilogue
          if constexpr (PREC == BEVR_PREC_F16) ds *= f32x2{c2_16, c2_16};
          s[r] = pp[0]; s[r + 1] = pp[1];
          dp[r] = ds[0]; dp[r + 1] = ds[1];
          sa2 += ds * (d0 + fx2 * (d1 - d0));   // d bias / d a
          sb2 += ds * du;                       // d bias / d b
        }
      }
      const float sa = sa2[0] + sa2[1], sb = sb2[0] + sb2[1];
      da[w] += sa;
      db[w] += sb;
#if BEVR_DROP
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = ((kmask >> r) & 1u) ? s[r] * ksc : 0.f;
#endif
      PROF_TD(t3, sa + sb + s[15]);
      PROF_ADD(2, t3 - t2);
      {
        Frag<PREC> dotf;
        load_perm(dotf, base + 2 * L::TILE_Q + L::TILE_T + lq * L::TSTRIDE + sub * 32 * EB, hi);
        dv[w] = mma_acc_b(dotf, s, dv[w]);
      }
      {
        Frag<PREC> qtf;
        load_perm(qtf, base + 2 * L::TILE_Q + lq * L::TSTRIDE + sub * 32 * EB, hi);
        dk[w] = mma_acc_b(qtf, dp, dk[w]);
      }
      PROF_TD(t4, dk[w][0] + dv[w][0]);
      PROF_ADD(3, t4 - t3);
    }

    }   // sub-tiles

    PROF_TD(t5, 0.f);
    if (pf_units > 0) {
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int u = wave + k * (TW / 64);
        const int c = u / n_chunk, ch = u - c * n_chunk;
        const int row = ch * 64 + lane;
        if (u < pf_units && row < sl.rows) win[slot_of(pf_first + c) * sl.pitch + row] = pf_v[k];
      }
    }
    if (it + 1 < n_it) qs.store(tid, smem + (buf ^ 1) * L::BUF);
    PROF_TD(t6, 0.f);
    __syncthreads();
    PROF_TD(t7, 0.f);
    PROF_ADD(4, t6 - t5);
    PROF_ADD(5, t7 - t6);
    PROF_ADD(6, t7 - t0);
    PROF_ADD(7, 1);
  }
#ifdef BEVR_PROF
  if (lane == 0 && (wave == 0 || wave == 11)) {
    for (int i = 0; i < 8; ++i) atomicAdd(&bevr_prof_k[(wave ? 8 : 0) + i], pacc[i]);
  }
#endif

  // ---- epilogue -------------------------------------------------------------------------------------
#pragma unroll
  for (int w = 0; w < KWW; ++w) {
    if (!wave_live[w]) continue;
    float* kr = dK + ((size_t)ph * d.Np + key[w]) * 32;
    float* vr = dV + ((size_t)ph * d.Np + key[w]) * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 a, b;
#pragma unroll
      for (int k = 0; k < 4; ++k) { a[k] = BEVR_LN2 * ds_inv * dk[w][4 * g4 + k]; b[k] = p_inv * dv[w][4 * g4 + k]; }
      *reinterpret_cast<f32x4*>(kr + 8 * g4 + 4 * hi) = a;
      *reinterpret_cast<f32x4*>(vr + 8 * g4 + 4 * hi) = b;
    }
    const float sa = BEVR_LN2 * ds_inv * (da[w] + __shfl_xor(da[w], 32));
    const float sb = BEVR_LN2 * ds_inv * (db[w] + __shfl_xor(db[w], 32));
    if (hi == 0) {
      atomicAdd(dkey_a + (size_t)(prob * d.groups + grp) * d.Np + key[w], sa);
      atomicAdd(dkey_b + (size_t)(prob * d.groups + grp) * d.Np + key[w], sb);
    }
  }
}

// =========================================================================================================
// Gather kernel (general path)
// =========================================================================================================
constexpr int TG = 256;   // 4 waves x 3 x 32 keys
constexpr int KW = 3;
constexpr int GSPLIT = 4;  // workgroups per key block: the few blocks on this path would otherwise run ~6 ms each, alone

template <int PREC>
__global__ __launch_bounds__(TG) void attn_bwd_k_gather_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ Qt, const char* __restrict__ K,
    const char* __restrict__ V, const float* __restrict__ key_a, const float* __restrict__ key_b,
    const char* __restrict__ table_pair, const char* __restrict__ dO, const char* __restrict__ dOt,
    const float* __restrict__ LSE, const float* __restrict__ delta, const float* __restrict__ grad_scale,
    float* __restrict__ dK, float* __restrict__ dV, float* __restrict__ dkey_a, float* __restrict__ dkey_b BEVR_DROP_PARAMS) {
  typedef LdsK<PREC> L;
  const float kp16 = PREC == BEVR_PREC_F16 ? grad_scale[2] : 0.f, c2_16 = PREC == BEVR_PREC_F16 ? grad_scale[3] : 1.f;
  const float ds_inv = PREC == BEVR_PREC_F16 ? grad_scale[4] : 1.f, p_inv = PREC == BEVR_PREC_F16 ? grad_scale[5] : 1.f;
  constexpr int EB = L::EB;
  __shared__ __attribute__((aligned(16))) char smem[2 * L::BUF];
  __shared__ __attribute__((aligned(16))) KBox red[4];

  const int n_kb = (d.Np + KEYS_WG - 1) / KEYS_WG;
  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / (n_kb * GSPLIT)) * 8 + xcd;
  if (ph >= n_ph) return;
  const int kblk = (slot % (n_kb * GSPLIT)) / GSPLIT;
  const int split = slot % GSPLIT;   // this workgroup's share of the query sweep
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int grp = hd / (d.heads / d.groups);
  const int qb = prob / d.q_div;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const int Mp = d.S * d.Sp;
  const int n_rb = d.Sp / 32;

  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = V + ((size_t)ph * d.Np) * 32 * EB;
  const float* ka = key_a + (size_t)(prob * d.groups + grp) * d.Np;
  const float* kb = key_b + (size_t)(prob * d.groups + grp) * d.Np;
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const int Hp8 = d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));

  // ---- this wave's keys: operands and table coordinates stay in registers for the whole sweep -------
  KeyC kc[KW];
  bool wave_live[KW];
  int key_idx[KW];
#pragma unroll
  for (int w = 0; w < KW; ++w) {
    const int k0 = kblk * KEYS_WG + (wave * KW + w) * 32;
    wave_live[w] = k0 < d.Np;                       // wave-uniform
    const int key = wave_live[w] ? k0 + lq : lq;    // dead sub-tiles read a valid address, never store
    key_idx[w] = key;
    kc[w] = make_keyc(ka[key], kb[key], d);
  }
  KBox box;
  {
    // the unit's box decides the owner: the same reduction over the same keys as the window kernel
    int a_lo = 0x7fffffff, a_hi = (int)0x80000000;
    float b_lo = 3.0e38f, b_hi = -3.0e38f;
#pragma unroll
    for (int w = 0; w < KW; ++w) {
      if (wave_live[w] && key_idx[w] < d.N) {
        int A;
        float fy, bc;
        key_split(ka[key_idx[w]], kb[key_idx[w]], d, A, fy, bc);
        a_lo = min(a_lo, A); a_hi = max(a_hi, A);
        b_lo = fminf(b_lo, bc); b_hi = fmaxf(b_hi, bc);
      }
    }
    box = wg_key_box(a_lo, a_hi, b_lo, b_hi, red, TG / 64, wave, lane);
  }
  if (make_slab(box, d, L::WCAP, ceilf(rx)).fits) return;   // the window kernel owns this block (uniform exit)

  Frag<PREC> kf[KW], vf[KW];
#pragma unroll
  for (int w = 0; w < KW; ++w) {
    kf[w].load(Kh + (size_t)key_idx[w] * 32 * EB, hi);
    vf[w].load(Vh + (size_t)key_idx[w] * 32 * EB, hi);
  }
  f32x16 dk[KW], dv[KW];
  float da[KW], db[KW];
#pragma unroll
  for (int w = 0; w < KW; ++w) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[w][r] = 0.f; dv[w][r] = 0.f; }
    da[w] = 0.f;
    db[w] = 0.f;
  }

  QStage<PREC, TG> qs;
  qs.init(tid, Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB, dO + ((size_t)ph * Mp) * 32 * EB,
          Qt + ((size_t)(qb * d.heads + hd) * 32) * Mp * EB, dOt + ((size_t)ph * 32) * Mp * EB,
          LSE + (size_t)ph * Mp, delta + (size_t)ph * Mp, Mp, kp16);
  const int n_all = d.S * n_rb;
  const int chunk = (n_all + GSPLIT - 1) / GSPLIT;
  const int it_first = split * chunk, n_it = min(n_all, it_first + chunk);
  if (it_first >= n_it) return;   // uniform
  qs.load(tid, (size_t)(it_first / n_rb) * d.Sp + (it_first % n_rb) * 32);
  qs.store(tid, smem);
  __syncthreads();

  for (int it = it_first; it < n_it; ++it) {
    const int buf = (it - it_first) & 1;
    const char* base = smem + buf * L::BUF;
    const int j = it / n_rb, i0 = (it % n_rb) * 32;
    if (it + 1 < n_it) {
      const int jn = (it + 1) / n_rb, rbn = (it + 1) % n_rb;
      qs.load(tid, (size_t)jn * d.Sp + rbn * 32);
    }
    const float jr = (float)j * rx;

    Frag<PREC> qf, dof, qtf, dotf;
    qf.load(base + lq * L::STRIDE, hi);
    dof.load(base + L::TILE + lq * L::STRIDE, hi);
    load_perm(qtf, base + 2 * L::TILE + lq * L::STRIDE, hi);
    load_perm(dotf, base + 3 * L::TILE + lq * L::STRIDE, hi);
    const f32x4* rc = reinterpret_cast<const f32x4*>(base + 4 * L::TILE);

#pragma unroll
    for (int w = 0; w < KW; ++w) {
      if (!wave_live[w]) continue;
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {   // rows 8 g4 + 4 hi + 0..3
        f32x4 l4 = rc[2 * g4 + hi], d4 = rc[8 + 2 * g4 + hi];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[4 * g4 + k] = l4[k]; dp[4 * g4 + k] = d4[k]; }
      }
      s = mma_frag(qf, kf[w], s);      // S[q][key] - LSE[q]
      dp = mma_frag(dof, vf[w], dp);   // dP[q][key] - delta[q]
#if BEVR_DROP
      unsigned kmask = 0u;
      const float ksc = 65536.0f / (65536.0f - (float)drop_thr);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t hrow = bevr_drop_row(drop_seed, (uint32_t)ph, (uint32_t)(j * d.Sp + i0 + crow(r, hi)));
        const bool keep = bevr_drop_keep(hrow, (uint32_t)key_idx[w], drop_thr);
        const float nd = rc[8 + 2 * (r >> 2) + hi][r & 3];
        dp[r] = keep ? fmaf(ksc, dp[r] - nd, nd) : nd;
        kmask |= (keep ? 1u : 0u) << r;
      }
#endif

      const KeyC c = kc[w];
      float tx = jr + c.b;
      float xf = floorf(tx);
      float fx = tx - xf;
      const char* tp = tbl + (unsigned)((int)xf * Hp8 + c.aoff + (i0 + 4 * hi) * 8);
      const bool dead = key_idx[w] >= d.N;
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x2 t0 = *reinterpret_cast<const f32x2*>(tp + crow(r, 0) * 8);
        f32x2 t1 = *reinterpret_cast<const f32x2*>(tp + Hp8 + crow(r, 0) * 8);
        float u0 = t0[0] * c.wy0 + t0[1] * c.fy;
        float u1 = t1[0] * c.wy0 + t1[1] * c.fy;
        float sv = s[r] + u0 + fx * (u1 - u0);
        float p = dead ? 0.f : fast_exp2(sv);
        float ds = BEVR_LN2 * c2_16 * p * dp[r];
        s[r] = p;
        dp[r] = ds;
        float ga = (t0[1] - t0[0]) + fx * ((t1[1] - t1[0]) - (t0[1] - t0[0]));  // d bias / d a
        sa += ds * ga;
        sb += ds * (u1 - u0);                                                  // d bias / d b
      }
      da[w] += sa;
      db[w] += sb;
#if BEVR_DROP
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = ((kmask >> r) & 1u) ? s[r] * ksc : 0.f;
#endif
      dv[w] = mma_acc_b(dotf, s, dv[w]);
      dk[w] = mma_acc_b(qtf, dp, dk[w]);
    }

    if (it + 1 < n_it) qs.store(tid, smem + (buf ^ 1) * L::BUF);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------
#pragma unroll
  for (int w = 0; w < KW; ++w) {
    if (!wave_live[w]) continue;
    const int key = key_idx[w];
    float* kr = dK + ((size_t)ph * d.Np + key) * 32;
    float* vr = dV + ((size_t)ph * d.Np + key) * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // partial sums of this workgroup's share of the sweep (rows cleared by the window kernel)
        atomicAdd(kr + 8 * g4 + 4 * hi + k, ds_inv * dk[w][4 * g4 + k]);
        atomicAdd(vr + 8 * g4 + 4 * hi + k, p_inv * dv[w][4 * g4 + k]);
      }
    }
    float sa = ds_inv * (da[w] + __shfl_xor(da[w], 32));
    float sb = ds_inv * (db[w] + __shfl_xor(db[w], 32));
    if (hi == 0) {
      atomicAdd(dkey_a + (size_t)(prob * d.groups + grp) * d.Np + key, sa);
      atomicAdd(dkey_b + (size_t)(prob * d.groups + grp) * d.Np + key, sb);
    }
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* Qt, const void* K, const void* V, const float* key_a,
           const float* key_b, const float* table_pair, const void* dO, const void* dOt, const float* LSE,
           const float* delta, const float* gs, float* dK, float* dV, float* dka, float* dkb, hipStream_t st BEVR_DROP_PARAMS) {
  const int n_kb = (d.Np + KEYS_WG - 1) / KEYS_WG;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_kb;
  hipLaunchKernelGGL((attn_bwd_k_win_kernel<PREC>), dim3(grid), dim3(TW), 0, st, d, (const char*)Q, (const char*)Qt,
                     (const char*)K, (const char*)V, key_a, key_b, (const char*)table_pair, (const char*)dO,
                     (const char*)dOt, LSE, delta, gs, dK, dV, dka, dkb BEVR_DROP_ARGS);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_k_gather_kernel<PREC>), dim3(grid * GSPLIT), dim3(TG), 0, st, d, (const char*)Q,
                     (const char*)Qt, (const char*)K, (const char*)V, key_a, key_b, (const char*)table_pair,
                     (const char*)dO, (const char*)dOt, LSE, delta, gs, dK, dV, dka, dkb BEVR_DROP_ARGS);
  return (int)hipGetLastError();
}

}  // namespace

#if BEVR_DROP
extern "C" int bevr_attn_bwd_k_dropout(const bevr_attn_desc* d, const void* Q, const void* Qt, const void* K, const void* V,
                                       const float* key_a, const float* key_b, const float* table_pair, const void* dO,
                                       const void* dOt, const float* LSE, const float* delta, const float* grad_scale,
                                       float* dK, float* dV, float* dkey_a, float* dkey_b, unsigned drop_thr,
                                       unsigned drop_seed, void* stream) {
  if (drop_thr >= 65536u) return BEVR_E_SHAPE;
#else
extern "C" int bevr_attn_bwd_k(const bevr_attn_desc* d, const void* Q, const void* Qt, const void* K, const void* V,
                               const float* key_a, const float* key_b, const float* table_pair, const void* dO,
                               const void* dOt, const float* LSE, const float* delta, const float* grad_scale,
                               float* dK, float* dV, float* dkey_a, float* dkey_b, void* stream) {
#endif
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !Qt || !K || !V || !key_a || !key_b || !table_pair || !dO || !dOt || !LSE || !delta || !dK || !dV ||
      !dkey_a || !dkey_b || (d->precision == BEVR_PREC_F16 && !grad_scale))
    return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(Qt) || !bevr_aligned16(K) || !bevr_aligned16(V) || !bevr_aligned16(dO) ||
      !bevr_aligned16(dOt) || !bevr_aligned16(dK) || !bevr_aligned16(dV) || !bevr_aligned16(table_pair))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16>(*d, Q, Qt, K, V, key_a, key_b, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV,
                                  dkey_a, dkey_b, st BEVR_DROP_ARGS);
  if (d->precision == BEVR_PREC_F16)
    return launch<BEVR_PREC_F16>(*d, Q, Qt, K, V, key_a, key_b, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV,
                                 dkey_a, dkey_b, st BEVR_DROP_ARGS);
  if (d->precision == BEVR_PREC_BF16X3)
    return launch<BEVR_PREC_BF16X3>(*d, Q, Qt, K, V, key_a, key_b, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV,
                               dkey_a, dkey_b, st BEVR_DROP_ARGS);
  return launch<BEVR_PREC_F32>(*d, Q, Qt, K, V, key_a, key_b, table_pair, dO, dOt, LSE, delta, grad_scale, dK, dV,
                               dkey_a, dkey_b, st BEVR_DROP_ARGS);
}
