// Attention backward, key side: dK, dV and the gradients of the keys' table coordinates (a_n, b_n).
// Key-stationary: a workgroup owns 4 waves x KW x 32 keys and sweeps all query tiles, so every per-key sum
// stays in registers and nothing is reduced across workgroups.  Orientation here is "key on the lane":
// tiles are S[query][key], which makes P and dS the B operands of
//   dV^T[c][key] += dO^T[c][q] P[q][key]      dK^T[c][key] += Q^T[c][q] dS[q][key]
// without any transpose, and makes the key's table coordinates lane constants: the 16 registers of a
// tile are 16 consecutive-ish BEV rows of one column, so the bilinear taps are reached with immediate
// offsets from one per-lane base address.
// Row constants (-LSE[q], -delta[q]) are preloaded into the accumulators of S and dP.
#include "bevr_common.h"

namespace {

constexpr int THREADS = 256;
constexpr int QT = 32;  // queries per iteration (one 32-row block of one BEV column)

template <int PREC> struct LdsK {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int STRIDE = 32 * EB + 16;  // every tile is 32 rows x 32 elements
  static constexpr int TILE = 32 * STRIDE;
  static constexpr int BUF = 4 * TILE + 2 * QT * 4;  // Q, dO, Qt, dOt, lse, delta
};

template <int PREC, int KW>
__global__ __launch_bounds__(THREADS) void attn_bwd_k_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ Qt, const char* __restrict__ K,
    const char* __restrict__ V, const float* __restrict__ key_a, const float* __restrict__ key_b,
    const char* __restrict__ table_pair, const char* __restrict__ dO, const char* __restrict__ dOt,
    const float* __restrict__ LSE, const float* __restrict__ delta, float* __restrict__ dK, float* __restrict__ dV,
    float* __restrict__ dkey_a, float* __restrict__ dkey_b) {
  typedef LdsK<PREC> L;
  constexpr int EB = L::EB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int keys_wg = 4 * KW * 32;
  const int n_kb = (d.Np + keys_wg - 1) / keys_wg;
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

  const char* Qh = Q + ((size_t)(qb * d.heads + hd) * Mp) * 32 * EB;
  const char* Qth = Qt + ((size_t)(qb * d.heads + hd) * 32) * Mp * EB;
  const char* dOh = dO + ((size_t)ph * Mp) * 32 * EB;
  const char* dOth = dOt + ((size_t)ph * 32) * Mp * EB;
  const float* LSEh = LSE + (size_t)ph * Mp;
  const float* dlth = delta + (size_t)ph * Mp;
  const char* Kh = K + ((size_t)ph * d.Np) * 32 * EB;
  const char* Vh = V + ((size_t)ph * d.Np) * 32 * EB;
  const float* ka = key_a + (size_t)(prob * d.groups + grp) * d.Np;
  const float* kb = key_b + (size_t)(prob * d.groups + grp) * d.Np;
  const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
  const int Hp8 = d.Hp * 8;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));

  // ---- this wave's keys: operands and table coordinates stay in registers for the whole sweep -------
  Frag<PREC> kf[KW], vf[KW];
  KeyC kc[KW];
  bool wave_live[KW];
  int key_idx[KW];
#pragma unroll
  for (int w = 0; w < KW; ++w) {
    const int k0 = kblk * keys_wg + (wave * KW + w) * 32;
    wave_live[w] = k0 < d.Np;                       // wave-uniform
    const int key = wave_live[w] ? k0 + lq : lq;    // dead sub-tiles read a valid address, never store
    key_idx[w] = key;
    kf[w].load(Kh + (size_t)key * 32 * EB, hi);
    vf[w].load(Vh + (size_t)key * 32 * EB, hi);
    kc[w] = make_keyc(ka[key], kb[key], d);
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

  // ---- staging of one query tile: Q, dO rows; Qt, dOt transposed+permuted; lse, delta --------------
  constexpr int CHR = 32 * EB / 16;        // 16-B chunks per 32-element row
  constexpr int CH_ARR = 32 * CHR;         // chunks per tile (128 / 256)
  constexpr int NCH = 4 * CH_ARR / THREADS;  // 2 / 4
  u32x4 st[NCH];
  float st_c = 0.f;
  const int n_it = d.S * n_rb;

  auto stage_load = [&](int it) {
    const int j = it / n_rb, rb = it % n_rb;
    const size_t mq0 = (size_t)j * d.Sp + rb * 32;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int cid = tid + c * THREADS;
      const int arr = cid / CH_ARR, a = cid % CH_ARR;   // arr is wave-uniform
      const int row = a / CHR, cc = a % CHR;
      const char* src;
      if (arr == 0) src = Qh + (mq0 * 32) * EB + (size_t)a * 16;
      else if (arr == 1) src = dOh + (mq0 * 32) * EB + (size_t)a * 16;
      else if (arr == 2) src = Qth + ((size_t)row * Mp + mq0) * EB + cc * 16;
      else src = dOth + ((size_t)row * Mp + mq0) * EB + cc * 16;
      st[c] = *reinterpret_cast<const u32x4*>(src);
    }
    if (tid < QT) st_c = LSEh[mq0 + tid];
    else if (tid < 2 * QT) st_c = dlth[mq0 + tid - QT];
  };
  auto stage_store = [&](int buf) {
    char* base = smem + buf * L::BUF;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int cid = tid + c * THREADS;
      const int arr = cid / CH_ARR, a = cid % CH_ARR;
      *reinterpret_cast<u32x4*>(base + arr * L::TILE + (a / CHR) * L::STRIDE + (a % CHR) * 16) = st[c];
    }
    if (tid < 2 * QT) *reinterpret_cast<float*>(base + 4 * L::TILE + tid * 4) = st_c;
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();

  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    const char* base = smem + buf * L::BUF;
    if (it + 1 < n_it) stage_load(it + 1);
    const int j = it / n_rb, i0 = (it % n_rb) * 32;
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
        for (int k = 0; k < 4; ++k) { s[4 * g4 + k] = -l4[k]; dp[4 * g4 + k] = -d4[k]; }
      }
      s = mma_frag(qf, kf[w], s);      // S[q][key] - LSE[q]
      dp = mma_frag(dof, vf[w], dp);   // dP[q][key] - delta[q]

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
        float ds = BEVR_LN2 * p * dp[r];
        s[r] = p;
        dp[r] = ds;
        float ga = (t0[1] - t0[0]) + fx * ((t1[1] - t1[0]) - (t0[1] - t0[0]));  // d bias / d a
        sa += ds * ga;
        sb += ds * (u1 - u0);                                                  // d bias / d b
      }
      da[w] += sa;
      db[w] += sb;
      dv[w] = mma_acc_b(dotf, s, dv[w]);
      dk[w] = mma_acc_b(qtf, dp, dk[w]);
    }

    if (it + 1 < n_it) stage_store(buf ^ 1);
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
      f32x4 a, b;
#pragma unroll
      for (int k = 0; k < 4; ++k) { a[k] = dk[w][4 * g4 + k]; b[k] = dv[w][4 * g4 + k]; }
      *reinterpret_cast<f32x4*>(kr + 8 * g4 + 4 * hi) = a;
      *reinterpret_cast<f32x4*>(vr + 8 * g4 + 4 * hi) = b;
    }
    float sa = da[w] + __shfl_xor(da[w], 32);
    float sb = db[w] + __shfl_xor(db[w], 32);
    if (hi == 0) {
      atomicAdd(dkey_a + (size_t)(prob * d.groups + grp) * d.Np + key, sa);
      atomicAdd(dkey_b + (size_t)(prob * d.groups + grp) * d.Np + key, sb);
    }
  }
}

template <int PREC, int KW>
int launch(const bevr_attn_desc& d, const void* Q, const void* Qt, const void* K, const void* V, const float* key_a,
           const float* key_b, const float* table_pair, const void* dO, const void* dOt, const float* LSE,
           const float* delta, float* dK, float* dV, float* dka, float* dkb, hipStream_t st) {
  const int keys_wg = 4 * KW * 32;
  const int n_kb = (d.Np + keys_wg - 1) / keys_wg;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * n_kb;
  const size_t lds = 2 * LdsK<PREC>::BUF;
  hipLaunchKernelGGL((attn_bwd_k_kernel<PREC, KW>), dim3(grid), dim3(THREADS), lds, st, d, (const char*)Q,
                     (const char*)Qt, (const char*)K, (const char*)V, key_a, key_b, (const char*)table_pair,
                     (const char*)dO, (const char*)dOt, LSE, delta, dK, dV, dka, dkb);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_attn_bwd_k(const bevr_attn_desc* d, const void* Q, const void* Qt, const void* K, const void* V,
                               const float* key_a, const float* key_b, const float* table_pair, const void* dO,
                               const void* dOt, const float* LSE, const float* delta, float* dK, float* dV,
                               float* dkey_a, float* dkey_b, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !Qt || !K || !V || !key_a || !key_b || !table_pair || !dO || !dOt || !LSE || !delta || !dK || !dV ||
      !dkey_a || !dkey_b)
    return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(Qt) || !bevr_aligned16(K) || !bevr_aligned16(V) || !bevr_aligned16(dO) ||
      !bevr_aligned16(dOt) || !bevr_aligned16(dK) || !bevr_aligned16(dV) || !bevr_aligned16(table_pair))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16, 2>(*d, Q, Qt, K, V, key_a, key_b, table_pair, dO, dOt, LSE, delta, dK, dV, dkey_a,
                                     dkey_b, st);
  return launch<BEVR_PREC_F32, 1>(*d, Q, Qt, K, V, key_a, key_b, table_pair, dO, dOt, LSE, delta, dK, dV, dkey_a,
                                  dkey_b, st);
}
