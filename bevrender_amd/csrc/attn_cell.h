// Cell-sorted attention kernels: the relative-position bias (and, in the backward, its table gradient) as a small
// matrix product on the matrix cores instead of a per-pair gather.
//
// Reference arithmetic (model/SCA_deform_attn.py:341-394): bias[i, j, n] = bilinear(rpe_table, ty = i + a_n,
// tx = j rx + b_n).  For ONE BEV column j and a 32-key tile whose keys all sit in the same few table cells,
//   bias[n][i] = sum_{k'} W[n][k'] * Tsh[k'][i],      k' = (c, r) over a CHUNK of 4 table columns x 4 table rows,
//   Tsh[(c, r)][i] = T[x0 + c][A0 + r + i]            the table, shifted by the query's BEV row i (integer shift!),
//   W[n][(c, r)]   = hat(c - (tx_n - x0)) * hat(r - (a_n - A0)),   hat(u) = max(0, 1 - |u|)   (the bilinear weights:
//                    two non-zero columns x two non-zero rows per key),
// i.e. one 32x32x16 MFMA per (32 keys x 32 BEV rows) tile, against 16 key rows x (3 LDS reads + ~7 VALU) for the
// gather of attn_fwd.hip.  Tsh depends only on the chunk's origin (x0, A0) and the wave's 32 BEV rows: a wave keeps
// it in 4 registers for as long as consecutive tiles share the origin.  W depends only on (j, key): the waves of a
// workgroup are the row blocks of ONE column j, so W is built once per workgroup and tile and shared through LDS.
//
// When does a tile fit one chunk?  When its keys span < 4 table columns and < 4 table rows INCLUDING their second
// taps.  The projector pins every pillar point outside a camera's image to pixel (0, 0) (model/bev_cmr_proj.py:76):
// two thirds of a view's keys sit at the same reference position and differ only by the learned offset, +-5 table
// rows x +-2.5 table columns -- ~70 cells holding ~1000 keys each.  The caller sorts those keys by table cell per call
// (softmax is invariant to key order; ops.cell_order) and 99.9 % of the sorted 32-key tiles fit one chunk.  Tiles that
// do not fit take a per-pair gather from the table in global memory (correct for any key set, slow): the caller keeps
// scattered keys on the region kernels (attn_fwd.hip ...) and chains the two key segments through (O, LSE).
#pragma once
#include "attn_tile.h"

// 16-byte load through a pointer known to be global memory (a pointer that went through a lambda capture or a select
// between kernel arguments is otherwise loaded with flat_load, which also counts on the LDS counter)
__device__ __forceinline__ u32x4 gload16(const char* p) {
  typedef const u32x4 __attribute__((address_space(1)))* gptr;
  return *(gptr)(unsigned long long)p;
}

constexpr int CELL_C = 4;   // table columns per chunk
constexpr int CELL_R = 4;   // table rows per chunk

// One chunk's operand of the bias product, per lane.  The MFMA A and B lane maps are symmetric (lane l holds
// [row or column l & 31][k = 8 (l >> 5) + e] in bf16, [..][k = l >> 5] per instruction in f32), so the same registers serve
// as A (lane = key, forward orientation S^T[key][query]) or as B (key-side backward, S[query][key]).
//   bf16: element e of lane half h  <->  cell k' = 8 h + e      = (column 2 h + (e >> 2), row e & 3)
//   f32 : step t    of lane half h  <->  cell k' = 2 t + h      = (column t >> 1,         row 2 (t & 1) + h)
template <int PREC> struct CellFrag { bf16x8 v; };                 // the 16-bit operand modes (raw bits)
template <> struct CellFrag<BEVR_PREC_F32> { float v[8]; };
// split mode: the same container as raw bits, v[0..3] = hi plane (8 bf16, element t <-> f32 step t), v[4..7] = lo plane
template <> struct CellFrag<BEVR_PREC_BF16X3> : CellFrag<BEVR_PREC_F32> {};
__device__ __forceinline__ void cell_split(CellFrag<BEVR_PREC_BF16X3>& f) {
  const Split8 sp = split8(f.v);
  put8(f.v, sp.hi);
  put8(f.v + 4, sp.lo);
}
template <int PREC> __device__ __forceinline__ void cell_split(CellFrag<PREC>&) {}

template <int PREC>
__device__ __forceinline__ f32x16 mma_cell(const CellFrag<PREC>& a, const CellFrag<PREC>& b, f32x16 acc) {
  return Half<PREC>::mfma(a.v, b.v, acc);
}
__device__ __forceinline__ f32x16 mma_cell(const CellFrag<BEVR_PREC_F32>& a, const CellFrag<BEVR_PREC_F32>& b, f32x16 acc) {
#pragma unroll
  for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 mma_cell(const CellFrag<BEVR_PREC_BF16X3>& a, const CellFrag<BEVR_PREC_BF16X3>& b, f32x16 acc) {
  return mma_split(Split8{raw8(a.v), raw8(a.v + 4)}, Split8{raw8(b.v), raw8(b.v + 4)}, acc);
}

__device__ __forceinline__ float hat(float u) { return fmaxf(1.0f - fabsf(u), 0.f); }

// Bilinear weights of this lane's key over the chunk's cells.  tcol = tx - x0 (column coordinate relative to chunk
// column 0: integer part = first tap column, fraction = fx), trow = a - A0 likewise.  A masked key passes tcol = -8.
template <int PREC> __device__ __forceinline__ CellFrag<PREC> cell_weights(float tcol, float trow, int h) {
  if constexpr (!is16(PREC)) {
    const float wy0 = hat((float)h - trow), wy1 = hat((float)(2 + h) - trow);
    CellFrag<PREC> f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float wx = hat((float)c - tcol);
      f.v[2 * c] = wx * wy0;
      f.v[2 * c + 1] = wx * wy1;
    }
    cell_split(f);
    return f;
  } else {
  const float wx0 = hat((float)(2 * h) - tcol), wx1 = hat((float)(2 * h + 1) - tcol);
  float wy[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wy[r] = hat((float)r - trow);
  u32x4 w;
  w[0] = Half<PREC>::pack2(wx0 * wy[0], wx0 * wy[1]);
  w[1] = Half<PREC>::pack2(wx0 * wy[2], wx0 * wy[3]);
  w[2] = Half<PREC>::pack2(wx1 * wy[0], wx1 * wy[1]);
  w[3] = Half<PREC>::pack2(wx1 * wy[2], wx1 * wy[3]);
  CellFrag<PREC> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
  }
}

// The table side of a chunk for this lane's BEV row: chunk column 0 = table column xc0, chunk row 0 for this lane =
// table row yr0 (both un-padded coordinates; yr0 already includes the lane's BEV row).  Reads are clamped into the
// padded table: a cell outside it is never given a non-zero weight (the keys' coordinates are clamped so that their
// taps stay inside, attn_keyprep.hip), so what a clamped read returns is irrelevant as long as it is finite.
template <int PREC>
__device__ __forceinline__ CellFrag<PREC> cell_table(const char* tbl, const bevr_attn_desc& d, int xc0, int yr0, int h) {
  if constexpr (!is16(PREC)) {
    const int ea = max(0, min(yr0 + h + d.y_off, d.Hp - 1)), eb = max(0, min(yr0 + h + 2 + d.y_off, d.Hp - 1));
    CellFrag<PREC> f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int xc = max(0, min(xc0 + c + d.x_off, d.Wp - 1));
      const char* col = tbl + (size_t)xc * d.Hp * 8;
      f.v[2 * c] = *reinterpret_cast<const float*>(col + (size_t)ea * 8);
      f.v[2 * c + 1] = *reinterpret_cast<const float*>(col + (size_t)eb * 8);
    }
    cell_split(f);
    return f;
  } else {
  const int e0 = max(0, min(yr0 + d.y_off, d.Hp - 1)), e2 = max(0, min(yr0 + d.y_off + 2, d.Hp - 1));
  u32x4 w;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int xc = max(0, min(xc0 + 2 * h + cc + d.x_off, d.Wp - 1));
    const char* col = tbl + (size_t)xc * d.Hp * 8;
    const f32x2 p0 = *reinterpret_cast<const f32x2*>(col + (size_t)e0 * 8);   // (T[y], T[y + 1])
    const f32x2 p2 = *reinterpret_cast<const f32x2*>(col + (size_t)e2 * 8);   // (T[y + 2], T[y + 3])
    w[2 * cc] = Half<PREC>::pack2(p0[0], p0[1]);
    w[2 * cc + 1] = Half<PREC>::pack2(p2[0], p2[1]);
  }
  CellFrag<PREC> f;
  f.v = __builtin_bit_cast(bf16x8, w);
  return f;
  }
}

// Geometry of one 32-key tile for BEV column j (uniform over the workgroup: functions of the scalar-loaded StepBox).
struct CellTile {
  int live;   // the tile has at least one unmasked key
  int fast;   // its taps fit one chunk
  int x0;     // first tap column of the tile for this BEV column: floor(j rx + bmin)
  int a0;     // first tap row for BEV row 0: amin
};
__device__ __forceinline__ CellTile make_celltile(const StepBox& sb, float jrx) {
  CellTile t;
  t.live = sb.amax >= sb.amin;
  t.x0 = (int)floorf(jrx + sb.bmin);
  t.a0 = sb.amin;
  const int x1 = (int)floorf(jrx + sb.bmax) + 1;   // last tap column
  t.fast = t.live && (x1 - t.x0 < CELL_C) && (sb.amax + 1 - sb.amin < CELL_R);
  return t;
}

// Cell coordinates of a key inside the tile's chunk (lane = key), from its prepared record.
__device__ __forceinline__ void cell_coords(const KeyW& kw, float jrx, int x0, bool dead, float& tcol, float& trow) {
  const float tx = jrx + kw.b;
  const float xf = floorf(tx);
  tcol = dead ? -8.0f : (xf - (float)x0) + (tx - xf);
  trow = (float)(kw.arow8 >> 3) + kw.fy;   // arow8 >> 3 = floor(a) - amin of the key's 32-key half
}
