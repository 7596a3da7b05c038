// Query-tile staging of the key-stationary backward kernels (attn_bwd_k.hip, attn_cell_bwd_k.hip): per 32-query
// tile the Q and dO rows, their transposes (perm32 order over the packed query index) and the row constants
// (-LSE, -delta) go global -> registers -> LDS one iteration ahead.
#pragma once
#include "bevr_common.h"

// NT = 32-query tiles staged (and processed) per barrier
template <int PREC, int NT = 1> struct LdsK {
  static constexpr int EB = Elem<PREC>::bytes;
  static constexpr int QTILE = NT * 32;
  static constexpr int STRIDE = 32 * EB + 16;            // row-layout tiles (Q, dO): bytes per query row
  static constexpr int TSTRIDE = QTILE * EB + 16;        // transposed tiles (Qt, dOt): bytes per channel row
  static constexpr int TILE_Q = QTILE * STRIDE;
  static constexpr int TILE_T = 32 * TSTRIDE;
  static constexpr int TILE = TILE_Q;                     // NT = 1: all four tiles are 32 rows x 32 elements
  static constexpr int BUF = 2 * TILE_Q + 2 * TILE_T + 2 * QTILE * 4;  // Q, dO, Qt, dOt, lse, delta
  static constexpr int WCAP = is16(PREC) ? 30720 : 26624;   // ring capacity, f32 entries (one workgroup per CU)
};

// ---- query-tile staging shared by both kernels ----------------------------------------------------------
template <int PREC, int THREADS, int NT = 1> struct QStage {
  typedef LdsK<PREC, NT> L;
  static constexpr int EB = L::EB;
  static constexpr int QTILE = L::QTILE;
  static constexpr int CHR = 32 * EB / 16;            // 16-B chunks per 32-element query row (Q, dO)
  static constexpr int CHT = QTILE * EB / 16;         // 16-B chunks per channel row of QTILE queries (Qt, dOt)
  static constexpr int CH_ARR = QTILE * CHR;          // chunks per array: QTILE * CHR = 32 * CHT
  static constexpr int NCH = (4 * CH_ARR + THREADS - 1) / THREADS;    // chunks per thread (the last slot may be partial)
  u32x4 st[NCH];
  float st_c;
  const char* base[NCH];    // array base pointer (uniform per chunk slot: the array index is wave-uniform)
  unsigned off[NCH];        // per-thread byte offset inside the array for query 0
  int mul[NCH];             // bytes per query index step
  int dst[NCH];
  const float* cbase;       // wave 0: LSE row constants, wave 1: delta
  float c_add;              // added to the (negated) row constant: fp16 mode's P scale exponent on the LSE seeds

  __device__ __forceinline__ void init(int tid, const char* Qh, const char* dOh, const char* Qth, const char* dOth,
                                       const float* LSEh, const float* dlth, int Mp, float lse_add = 0.f) {
    static_assert(CH_ARR % 64 == 0, "a chunk slot's array must be wave-uniform");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int cid = tid + c * THREADS;
      const int arr = __builtin_amdgcn_readfirstlane(cid / CH_ARR), a = cid % CH_ARR;
      base[c] = arr == 0 ? Qh : arr == 1 ? dOh : arr == 2 ? Qth : dOth;   // arr >= 4: idle slot (dst < 0)
      if (arr < 2 || arr >= 4) {
        const int row = a / CHR, cc = a % CHR;
        off[c] = (unsigned)a * 16; mul[c] = 32 * EB;
        dst[c] = arr * L::TILE_Q + row * L::STRIDE + cc * 16;
      } else {
        const int row = a / CHT, cc = a % CHT;
        off[c] = (unsigned)(((size_t)row * Mp) * EB + cc * 16); mul[c] = EB;
        dst[c] = 2 * L::TILE_Q + (arr - 2) * L::TILE_T + row * L::TSTRIDE + cc * 16;
      }
      if (cid >= 4 * CH_ARR) dst[c] = -1;
    }
    st_c = 0.f;
    cbase = (tid >> 6) == 0 ? LSEh : dlth;
    c_add = (tid >> 6) == 0 ? lse_add : 0.f;
  }
  __device__ __forceinline__ void load(int tid, size_t mq0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (dst[c] >= 0) st[c] = *reinterpret_cast<const u32x4*>(base[c] + mq0 * mul[c] + off[c]);
    if (tid < 128 && (tid & 63) < QTILE) st_c = c_add - cbase[mq0 + (tid & 63)];   // negated: they seed the accumulators
  }
  __device__ __forceinline__ void store(int tid, char* buf) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (dst[c] >= 0) *reinterpret_cast<u32x4*>(buf + dst[c]) = st[c];
    if (tid < 128 && (tid & 63) < QTILE)
      *reinterpret_cast<float*>(buf + 2 * L::TILE_Q + 2 * L::TILE_T + ((tid >> 6) * QTILE + (tid & 63)) * 4) = st_c;
  }
};

