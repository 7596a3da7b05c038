// Query-stationary tile machinery shared by the forward and the query-side backward:
// per 64-key step, the keys' table coordinates, the bounding box of the table taps they need for this
// workgroup's query tile, and the LDS window holding exactly that part of the rpe table.
//
// Why a window: a (query tile) x (key step) block touches the table only inside
//   rows    [i0 + min floor(a), i0 + 31 + max floor(a) + 1]
//   columns [floor(j_lo rx + min b), floor(j_hi rx + max b) + 1]
// When the keys of a step are spatially compact (regular TSA grid; SCA keys ordered along a k-d tree of
// their static camera projections) that box is a few thousand entries, so the 4-tap bilinear gather of
// every (query, key) pair is served from LDS -- conflict-free, because the 32 lanes of a tile are 32
// consecutive table rows of one column -- instead of from L2.  Steps whose box does not fit fall back to
// gathering from global memory; both paths compute the same thing.
#pragma once
#include "bevr_common.h"

constexpr int KT = 64;           // keys per step
constexpr int THREADS = 256;     // 4 waves
constexpr int WIN_PITCH = 64;    // window rows (pair entries) per column = one wave-wide load per column
constexpr int WIN_ROWS_MAX = 63; // 32 + (Amax - Amin) must not exceed this (accumulation window needs +1 row)

// per-key constants in LDS (16 B, read as a broadcast)
struct KeyW {
  int aoff;    // global path: byte offset ((A + y_off) + x_off * Hp) * 8 into the head's pair table
  float fy;    // frac(a)
  float b;     // clamped column coordinate
  int arow8;   // window path: (A - Amin of the key's 32-key half) * 8 + the key's group id within the half (0..7)
};

// Column capacity of the table regions, per precision mode and kernel (the LDS budgets are in the kernels).
// The key-preparation kernel sizes the GROUPS of a half (below) for the smaller of the two.
__host__ __device__ constexpr int region_cap_fwd(int prec) { return 88; }
__host__ __device__ constexpr int region_cap_bwd_q(int prec) { return is16(prec) ? 56 : 48; }
__host__ __device__ constexpr int region_cap_min(int prec) {
  return region_cap_fwd(prec) < region_cap_bwd_q(prec) ? region_cap_fwd(prec) : region_cap_bwd_q(prec);
}
constexpr int QCOLS = 8;        // query columns per workgroup of the query-stationary kernels (one per wave)
constexpr int N_GROUP = 8;      // groups per 32-key half: 2 row bands x 4 column bands
constexpr int GROUPS_NONE = 0x7ffffffe;   // gbox[0].amin of a half that has no groups (distinct from an empty group's INT_MAX)

struct WinInfo {
  int ok;       // window path valid for this step
  int xlo;      // first table column in the window (un-padded coordinates)
  int ncols;
  int amin;     // min floor(a) over the live keys of the step
  int nrows;    // 32 + amax - amin
  float xlo_f;
  int pad0, pad1;
};

// bounding box of the live keys of one 32-key half of a step, written by the key-preparation kernel (attn_keyprep.hip)
struct StepBox {
  int amin, amax;      // min / max floor(a); amax < amin: the step has no live key
  float bmin, bmax;
};

// Byte layout of the key workspace handed between bevr_attn_key_prep and the query-stationary kernels:
//   KeyW   [n_prob * groups][Np]                  then
//   StepBox[n_prob * groups][Np / 32]             one box per 32-key half of a step, then
//   StepBox[n_prob * groups][Np / 32][N_GROUP]    the boxes of the half's groups
// Groups: a half whose box fits no region (its 32 keys are too far apart in table space: sparse far-field keys) is
// not sent down the global-memory path any more (per-pair global gathers / float atomics: 1.5 % of the halves at cfg2
// cost more than the other 98.5 % together).  Its keys are banded -- row band = (A - Amin) / 32, column band =
// (b - bmin) / group_width -- and the kernels run one windowed pass per non-empty group with the other keys masked
// (kill column).  Each group's box fits every region by construction (group_width below); a half spread over more
// than 2 row bands or 4 column bands has no groups (gbox[0].amin = GROUPS_NONE marks it) and takes the global path.
__host__ __device__ __forceinline__ size_t key_ws_box_offset(const bevr_attn_desc& d) {
  return (size_t)d.n_prob * d.groups * d.Np * sizeof(KeyW);
}
__host__ __device__ __forceinline__ size_t key_ws_gbox_offset(const bevr_attn_desc& d) {
  return key_ws_box_offset(d) + (size_t)d.n_prob * d.groups * (d.Np / 32) * 16;
}
// widest spread of b a group may have so that its window fits `cap` columns for every 8-column query tile:
// ncols = floor(jhi + bmax) + 2 - (floor(jlo + bmin) - 1) + 1 <= (QCOLS - 1) rx + (bmax - bmin) + 5
__host__ __device__ __forceinline__ float group_width(const bevr_attn_desc& d) {
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  return (float)region_cap_min(d.precision) - (float)(QCOLS - 1) * rx - 5.5f;
}

// Union of the two halves' boxes of a step (an empty half has amax < amin and huge bmin / -huge bmax: neutral).
__device__ __forceinline__ StepBox box_union(const StepBox& a, const StepBox& b) {
  StepBox u;
  u.amin = min(a.amin, b.amin);
  u.amax = max(a.amax, b.amax);
  u.bmin = fminf(a.bmin, b.bmin);
  u.bmax = fmaxf(a.bmax, b.bmax);
  return u;
}

// The table box a (query tile) x (key step) block needs, from the step's key box and the tile's column range.
// Uniform over the workgroup; every thread evaluates it (a dozen ALU operations per step).
__device__ __forceinline__ WinInfo make_wininfo(const StepBox& sb, float jrx_lo, float jrx_hi, int max_cols) {
  WinInfo wi;
  wi.amin = sb.amin;
  wi.nrows = 32 + sb.amax - sb.amin;
  wi.xlo = (int)floorf(jrx_lo + sb.bmin) - 1;
  const int xhi = (int)floorf(jrx_hi + sb.bmax) + 2;
  wi.ncols = xhi - wi.xlo + 1;
  wi.xlo_f = (float)wi.xlo;
  wi.ok = (sb.amax >= sb.amin) && (wi.nrows <= WIN_ROWS_MAX) && (wi.ncols <= max_cols);
  wi.pad0 = wi.pad1 = 0;
  return wi;
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// per-(column, key) constants, written by the owning wave once per step (lane = key), read as a broadcast.
// bf16 mode keeps only the packed bf16 tap weights (the gradient unpacks them: the weights the bias was computed
// with); f32 mode keeps them in full precision.
template <int PREC> struct ColKeyT {   // the 16-bit operand modes (bf16, fp16); f32 is specialised below
  // member order: the weights are one aligned 8-byte read (ds_read_b64, 2 clk) and the cell one 4-byte read; with the
  // cell first the compiler read the weights with ds_read2_b32 (4 clk)
  unsigned wA, wB;  // tap weights of column x / x + 1 as a packed 16-bit pair (row y, row y + 1)
  int cell;         // first tap's window position for lane row 0: cell index (backward) or byte offset (forward)
  unsigned pad;
  __device__ __forceinline__ void set(float w00, float w01, float w10, float w11) {
    wA = Half<PREC>::pack2(w00, w01);
    wB = Half<PREC>::pack2(w10, w11);
    pad = 0;
  }
  __device__ __forceinline__ float w00() const { return Half<PREC>::lo(wA); }
  __device__ __forceinline__ float w01() const { return Half<PREC>::hi(wA); }
  __device__ __forceinline__ float w10() const { return Half<PREC>::lo(wB); }
  __device__ __forceinline__ float w11() const { return Half<PREC>::hi(wB); }
};
template <> struct ColKeyT<BEVR_PREC_F32> {
  int cell;
  unsigned pad0, pad1, pad2;
  float f00, f01, f10, f11;   // (1-fx)(1-fy), (1-fx)fy, fx(1-fy), fx fy
  __device__ __forceinline__ void set(float w00, float w01, float w10, float w11) {
    f00 = w00; f01 = w01; f10 = w10; f11 = w11;
    pad0 = pad1 = pad2 = 0;
  }
  __device__ __forceinline__ float w00() const { return f00; }
  __device__ __forceinline__ float w01() const { return f01; }
  __device__ __forceinline__ float w10() const { return f10; }
  __device__ __forceinline__ float w11() const { return f11; }
};
template <> struct ColKeyT<BEVR_PREC_BF16X3> : ColKeyT<BEVR_PREC_F32> {};

// ---------------------------------------------------------------------------------------------------
// Persistent window ("region"): a fixed-capacity box of the table, WIN_PITCH rows x NCOL columns, anchored
// at table coordinates (ax0, ay0) relative to this workgroup's first BEV row.  A step whose bounding box
// lies inside the current region reuses it; otherwise the region is re-anchored (centred on the new box) and
// refilled.  Consecutive steps of a k-d ordered key list mostly stay inside one region -- in particular the
// long runs of out-of-image keys that the projector pins to one pixel.
struct Region {
  int ax0, ay0;   // un-padded table column of window column 0; floor(a) value of window row 0 (for lane row 0)
};

__device__ __forceinline__ bool region_contains(const Region& rg, const WinInfo& wi, int ncol_cap) {
  return wi.xlo >= rg.ax0 && wi.xlo + wi.ncols <= rg.ax0 + ncol_cap && wi.amin >= rg.ay0 &&
         wi.amin + wi.nrows <= rg.ay0 + WIN_ROWS_MAX;
}

// Centre the region on the step's box, clamped so that every window entry is inside the padded table.
__device__ __forceinline__ Region region_anchor(const WinInfo& wi, const bevr_attn_desc& d, int i0, int ncol_cap) {
  Region rg;
  int xi = wi.xlo - (ncol_cap - wi.ncols) / 2 + d.x_off;                 // padded column index of window col 0
  xi = max(0, min(xi, d.Wp - ncol_cap));
  int yi = i0 + wi.amin - (WIN_ROWS_MAX - wi.nrows) / 2 + d.y_off;       // padded row index of window row 0
  yi = max(0, min(yi, d.Hp - WIN_PITCH));
  rg.ax0 = xi - d.x_off;
  rg.ay0 = yi - d.y_off - i0;
  return rg;
}

// One (T[y], T[y+1]) entry of region column c for this lane's row, for the region fill (lane = row).
// The region has a fixed capacity; the padded table of a SMALL problem (S = 8: Wp = 38) is narrower than that, and
// region_anchor can then only clamp column 0 of the region to column 0 of the table: region columns past the
// table's last column exist in LDS but must not be fetched -- no key's tap can point at them (taps are clamped into
// the padded table), so they are filled with zeros.  Round 1 fetched them: up to 50 columns x Hp x 8 B past the end
// of the last head's table, a read that faults only when the allocation happens to end at an unmapped page (the
// "Fatal Python error: Aborted" of gpurun_out/gpu_tests_6.log, and again of r02_gpu_2.log: both in the S = 8 TSA
// module test, the first launch with a 15-column table).
__device__ __forceinline__ f32x2 region_entry(const char* tbl, const bevr_attn_desc& d, const Region& rg, int c, size_t y0) {
  const int xc = rg.ax0 + c + d.x_off;   // padded table column: >= 0 by region_anchor
  f32x2 v = {0.f, 0.f};
  if (xc < d.Wp) v = *reinterpret_cast<const f32x2*>(tbl + ((size_t)xc * d.Hp + y0) * 8);
  return v;
}

// Workgroup-uniformity contract of the query-stationary kernels: every __syncthreads() they execute conditionally
// (region moves, mid-step moves) sits under predicates computed ONLY from kernel arguments, blockIdx and the
// scalar-loaded StepBox records -- never from threadIdx, the wave's column or a key's data.  -DBEVR_DEBUG builds
// check it: thread 0 publishes the predicate, every thread compares, a mismatch traps instead of hanging.
#ifdef BEVR_DEBUG
#define BEVR_ASSERT_WG_UNIFORM(EXPR_)                                               \
  do {                                                                              \
    __shared__ int bevr_dbg_u;                                                      \
    __syncthreads();                                                                \
    if (threadIdx.x == 0) bevr_dbg_u = (int)(EXPR_);                                \
    __syncthreads();                                                                \
    if (bevr_dbg_u != (int)(EXPR_)) __builtin_trap();                               \
    __syncthreads();                                                                \
  } while (0)
#define BEVR_ASSERT(EXPR_) do { if (!(EXPR_)) __builtin_trap(); } while (0)
#else
#define BEVR_ASSERT_WG_UNIFORM(EXPR_) do { } while (0)
#define BEVR_ASSERT(EXPR_) do { } while (0)
#endif
