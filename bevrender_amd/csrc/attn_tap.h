// TAP kernels: attention over the key segment the projector PINS to feature pixel (0, 0), without K and V.
//
// Reference arithmetic (model/SCA_deform_attn.py:290-321, 331-413; model/bev_cmr_proj.py:76).  A pillar point outside a
// camera's image is projected to the normalised reference (-1, -1) = feature pixel (0, 0); its sampling position
// differs from that pixel only by the learned offset, tanh(.) * 5 / (Hk - 1) in y and 5 / (Wk - 1) in x, i.e. at most
// +-2.5 (Hi - 1) / (Hk - 1) x +-2.5 (Wi - 1) / (Wk - 1) feature pixels (+-1.6 x +-0.44 at the benchmark shape).  With
// zero padding outside the image EVERY such key samples inside the top-left TAP_R x TAP_C pixels f_t:
//     x_s(n) = sum_t w_t(n) f_t,      w_t(n) = hat(r_t - ys_n) hat(c_t - xs_n)        (bilinear weights, hat(u) = max(0, 1 - |u|))
//     K_n = Wk x_s(n) + bk = sum_t w_t(n) Kpix_t + bk,     V_n likewise        (proj_k / proj_v are 1x1 convolutions: linear)
// so for one (problem, head)
//     S[n][i]  = scale Q_i . K_n + bias = sum_t w_t(n) G[t][i] + Gb[i] + bias[n][i],    G[t][i] = scale Kpix_t . Q_i
//     O_i      = sum_n P[n][i] V_n      = sum_t R[t][i] Vpix_t + l_i bv,                R[t][i] = sum_n w_t(n) P[n][i]
// G is a (12 x M) GEMM and O = R Vpix a (M x 12) x (12 x c) one, both left to the caller (rocBLAS / torch autograd); what
// remains per (query, key) pair is ONE contraction over [12 taps | 16 bias cells] (attn_cell.h: the bias of a cell-sorted
// 32-key tile is a product with the 4 x 4 table chunk shifted by the query's BEV row), the exponential, and
// R += w^T P -- no K / V tile is staged or read, and the head width never enters.
//
// Matrix shapes: v_mfma_f32_16x16x32 (contraction 32 = 16 tap slots + 16 cells).  A tile is 16 keys x 16 BEV rows:
//   S^T[key][row]  = A[key][k] B[k][row],       A = [w | Wc] (lane = key), B = [G ; Tsh] (lane = BEV row); the row's offset
//                                                Gb - reference rides in two slots of G (hi + lo parts) against ones in w
//   R[slot][row]  += w^T[slot][key] P[key][row]  the accumulator of S^T (rows = keys) is the B operand as it stands; w^T
//                                                comes out of the SAME LDS image as A through ds_read_b64_tr_b16
// Slot TAP_ONE of w is the constant 1: row TAP_ONE of R is the softmax denominator (no VALU row sum); slot TAP_DEAD is 1
// for a masked key and G[TAP_DEAD][.] = -big: a masked key's weight is exp2(-big) = 0 with no compare in the loop.
#pragma once
#include "attn_cell.h"

constexpr int TAP_R = 4;        // feature rows 0..3 and
constexpr int TAP_C = 3;        // columns 0..2: slot t = r * TAP_C + c  (ys < 3, xs < 2: the caller checks the offset range)
constexpr int TAP_N = TAP_R * TAP_C;
constexpr int TAP_SLOTS = 16;   // 12 taps, TAP_CHI, TAP_CLO, TAP_DEAD, TAP_ONE
constexpr int TAP_CHI = 12;     // key side 1; query side the hi and lo 16-bit parts of the row's logit offset (Gb - reference):
constexpr int TAP_CLO = 13;     //   the MFMA adds it, no accumulator start registers
constexpr int TAP_DEAD = 14;
constexpr int TAP_ONE = 15;
constexpr int QB = 16;          // BEV rows per matrix tile

typedef __attribute__((ext_vector_type(4))) short s16x4;

// per key, written by bevr_attn_tap_prep: clamped table coordinates and the sampling position in feature pixels
// (a masked key: ys = TAP_YS_DEAD, every tap weight 0)
struct TapRec { float a, b, ys, xs; };
#define TAP_YS_DEAD (-100.0f)

// workspace layout: TapRec[n_prob][Np] | StepBox[n_prob][Np / 32]
__host__ __device__ __forceinline__ size_t tap_ws_box_offset(const bevr_attn_desc& d) {
  return (size_t)d.n_prob * d.Np * sizeof(TapRec);
}

template <int PREC> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma16<BEVR_PREC_BF16>(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16<BEVR_PREC_F16>(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two transposed LDS reads (4 rows x 16 columns of 16-bit each, rows 32 B apart in a [key][16] image): element e of
// the result = image[row r0 + e (e < 4) | r1 + e - 4][column lane & 15].  `p` is this lane's address in the first block
// (row r0 + ((lane & 15) >> 2), byte 8 (lane & 3)), `off2` the byte distance to the second block.
__device__ __forceinline__ bf16x8 lds_tr8(const char* p, int off2) {
  typedef s16x4 __attribute__((address_space(3)))* lp;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + off2));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// the 16 tap slots of one key as 8 packed dwords
template <int PREC> __device__ __forceinline__ void tap_weights(float ys, float xs, u32x4& lo, u32x4& hi8) {
  float wy[TAP_R], wx[TAP_C];
#pragma unroll
  for (int r = 0; r < TAP_R; ++r) wy[r] = hat((float)r - ys);
#pragma unroll
  for (int c = 0; c < TAP_C; ++c) wx[c] = hat((float)c - xs);
  float w[TAP_SLOTS];
#pragma unroll
  for (int r = 0; r < TAP_R; ++r)
#pragma unroll
    for (int c = 0; c < TAP_C; ++c) w[r * TAP_C + c] = wy[r] * wx[c];
  w[TAP_CHI] = 1.0f;
  w[TAP_CLO] = 1.0f;
  w[TAP_DEAD] = ys < -50.0f ? 1.0f : 0.f;
  w[TAP_ONE] = 1.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lo[k] = Half<PREC>::pack2(w[2 * k], w[2 * k + 1]);
    hi8[k] = Half<PREC>::pack2(w[8 + 2 * k], w[8 + 2 * k + 1]);
  }
}

// the 16 cells of the bias chunk for one key (both lane halves of cell_weights: cells 0..7 | 8..15)
template <int PREC> __device__ __forceinline__ void tap_cells(float tcol, float trow, u32x4& c0, u32x4& c1) {
  c0 = __builtin_bit_cast(u32x4, cell_weights<PREC>(tcol, trow, 0).v);
  c1 = __builtin_bit_cast(u32x4, cell_weights<PREC>(tcol, trow, 1).v);
}

// reductions inside a 32-lane half
__device__ __forceinline__ int half_min_i(int v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = min(v, __shfl_xor(v, s));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// The key stream of the query-stationary tap kernels (forward, query-side backward): LDS layout and the producer wave.
struct LdsT {
  static constexpr int OFF_TAPS = 0;            // [64 keys][16 slots] 16-bit
  static constexpr int OFF_CELLS = 2048;        // [64 keys][16 cells] 16-bit
  static constexpr int OFF_CT = 4096;           // u32x4: flags (bit 0 / 1: tile 0 / 1 live, bit 2: done), alloc0, alloc1, 0
  static constexpr int OFF_ORG = 4096 + 16;     // i32x4: chunk origin of tile 0 (x0, a0), of tile 1 (x0, a0)
  static constexpr int BUF = 4096 + 32;
  static constexpr int RING = 4;
};

__device__ __forceinline__ bool box_fits(const StepBox& sb, float jrx) {
  const int x0 = (int)floorf(jrx + sb.bmin), x1 = (int)floorf(jrx + sb.bmax) + 1;
  return sb.amax >= sb.amin && (x1 - x0 < CELL_C) && (sb.amax + 1 - sb.amin < CELL_R);
}

// The producer wave of the query-stationary tap kernels.  Emits the key stream of (problem `prob`, column j) into the two
// LDS buffers at `smem` and the ring of table images behind them; one __syncthreads() per emission, a final one with the
// done flag.  rows_img = BEV rows covered by an image (16 x row blocks of the column).
template <int PREC>
__device__ __forceinline__ void tap_producer(const bevr_attn_desc& d, char* smem, char* ring, int img_bytes, int rows_img,
                                             const TapRec* __restrict__ recs, const StepBox* __restrict__ box,
                                             const char* __restrict__ tbl, float jrx, int lane) {
  typedef LdsT L;
  const int hi = lane >> 5;
  const int n_step = d.Np / KT;
  int alloc = 0, tag_x = 1 << 30, tag_a = 1 << 30;
  int e = 0;
  // the producer is the workgroup's pacemaker: every other wave waits for its emission at the barrier, and it shares
  // its SIMD with row-block waves that would otherwise take most of the issue slots
  __builtin_amdgcn_s_setprio(3);
  // the table side of chunk origin (x0, a0): image[row][cell 4 c + r] = T2[x0 + c][a0 + row + r], 16-bit
  auto build_image = [&](char* img, int x0, int a0) {
    for (int row = lane; row < rows_img; row += 64) {
      const int yr0 = a0 + row + d.y_off;
      const int e0 = max(0, min(yr0, d.Hp - 1)), e2 = max(0, min(yr0 + 2, d.Hp - 1));
      u32x4 w0, w1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int xc = max(0, min(x0 + c + d.x_off, d.Wp - 1));
        const char* col = tbl + (size_t)xc * d.Hp * 8;
        const f32x2 p0 = *reinterpret_cast<const f32x2*>(col + (size_t)e0 * 8);
        const f32x2 p2 = *reinterpret_cast<const f32x2*>(col + (size_t)e2 * 8);
        const uint32_t u0 = Half<PREC>::pack2(p0[0], p0[1]), u1 = Half<PREC>::pack2(p2[0], p2[1]);
        if (c < 2) { w0[2 * c] = u0; w0[2 * c + 1] = u1; }
        else { w1[2 * (c - 2)] = u0; w1[2 * (c - 2) + 1] = u1; }
      }
      *reinterpret_cast<u32x4*>(img + row * 32) = w0;
      *reinterpret_cast<u32x4*>(img + row * 32 + 16) = w1;
    }
  };
  // image 0 stands for "no chunk yet" (allocation numbers start at 1): finite values for the masked keys of a slot that
  // precedes every live tile
  for (int o = lane * 16; o < img_bytes; o += 64 * 16) *reinterpret_cast<u32x4*>(ring + o) = u32x4{0u, 0u, 0u, 0u};
  TapRec rc_n = recs[lane];
  StepBox sb_n = box[hi];
  for (int step = 0; step < n_step; ++step) {
    const TapRec rc = rc_n;
    const StepBox sb = sb_n;
    if (step + 1 < n_step) {   // the next step's records are in flight while this one is emitted
      rc_n = recs[(size_t)(step + 1) * KT + lane];
      sb_n = box[2 * (step + 1) + hi];
    }
    bool rem = rc.ys > -50.0f;                       // keys of this lane's tile not emitted yet
    if (__ballot(rem) == 0ull) continue;             // a step of padding only: nothing to emit
    const int A = (int)floorf(rc.a);
    const float tx = jrx + rc.b;
    const float xf = floorf(tx);
    const int X = (int)xf;
    bool whole = box_fits(sb, jrx);                  // uniform over the half: the tile's own box fits one chunk
    do {
      int a0, x0;
      bool sel;
      if (whole) {
        a0 = sb.amin;
        x0 = (int)floorf(jrx + sb.bmin);
        sel = rem;
      } else {
        // one chunk's worth of the remaining keys: rows from the lowest remaining key, columns from the lowest key
        // among those (that key is always selected: progress)
        a0 = half_min_i(rem ? A : 0x7fffffff);
        const bool rowok = rem && A < a0 + CELL_R - 1;
        x0 = half_min_i(rowok ? X : 0x7fffffff);
        sel = rowok && X < x0 + CELL_C - 1;
      }
      whole = false;
      const unsigned long long selm = __ballot(sel);
      const int ok0 = (selm & 0xffffffffull) != 0ull, ok1 = (selm >> 32) != 0ull;
      char* bb = smem + (e & 1) * L::BUF;
      {
        u32x4 t0, t1, c0, c1;
        tap_weights<PREC>(sel ? rc.ys : TAP_YS_DEAD, rc.xs, t0, t1);
        const float tcol = sel ? (xf - (float)x0) + (tx - xf) : -8.0f;
        tap_cells<PREC>(tcol, rc.a - (float)a0, c0, c1);
        *reinterpret_cast<u32x4*>(bb + L::OFF_TAPS + lane * 32) = t0;
        *reinterpret_cast<u32x4*>(bb + L::OFF_TAPS + lane * 32 + 16) = t1;
        *reinterpret_cast<u32x4*>(bb + L::OFF_CELLS + lane * 32) = c0;
        *reinterpret_cast<u32x4*>(bb + L::OFF_CELLS + lane * 32 + 16) = c1;
      }
      int al0 = alloc, al1 = alloc, ox0 = tag_x, oa0 = tag_a;
      if (ok0) {
        const int x = __builtin_amdgcn_readlane(x0, 0), a = __builtin_amdgcn_readlane(a0, 0);
        if (x != tag_x || a != tag_a) {
          ++alloc;
          build_image(ring + (alloc & (L::RING - 1)) * img_bytes, x, a);
          tag_x = x;
          tag_a = a;
        }
        al0 = alloc;
        ox0 = tag_x;
        oa0 = tag_a;
      }
      if (ok1) {
        const int x = __builtin_amdgcn_readlane(x0, 32), a = __builtin_amdgcn_readlane(a0, 32);
        if (x != tag_x || a != tag_a) {
          ++alloc;
          build_image(ring + (alloc & (L::RING - 1)) * img_bytes, x, a);
          tag_x = x;
          tag_a = a;
        }
        al1 = alloc;
      }
      if (lane == 0) {
        *reinterpret_cast<u32x4*>(bb + L::OFF_CT) = u32x4{(unsigned)(ok0 | (ok1 << 1)), (unsigned)al0, (unsigned)al1, 0u};
        *reinterpret_cast<u32x4*>(bb + L::OFF_ORG) = u32x4{(unsigned)ox0, (unsigned)oa0, (unsigned)tag_x, (unsigned)tag_a};
      }
      rem = rem && !sel;
      ++e;
      __syncthreads();
    } while (__ballot(rem) != 0ull);
  }
  if (lane == 0) *reinterpret_cast<u32x4*>(smem + (e & 1) * L::BUF + L::OFF_CT) = u32x4{4u, 0u, 0u, 0u};
  __syncthreads();
}

