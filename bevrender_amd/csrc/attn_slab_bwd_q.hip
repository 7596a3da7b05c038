// Attention backward, query side (dQ and the rpe-table gradient), SLAB-STATIONARY: bevr_attn_slab_bwd_q.
//
// Same arithmetic per (query, key) pair as attn_bwd_q.hip (reference model/SCA_deform_attn.py:331-413,
// model/TSA_deform_attn.py:245-333 differentiated):  S^T[key][query] tiles with the query on the lane, P = exp2(S - LSE),
// dS = P (dP - delta), dQ += dS K, and the table gradient of the pair's four bilinear taps as two pre-summed 64-bit
// fixed-point LDS adds per key row -- but the work is cut along the TABLE instead of along the queries:
//
//   a work item owns a SLAB of `sw` consecutive table columns [x0, x0 + sw) of one (problem, head) -- ALL rows of them: the
//   table values (16-bit pairs) and the gradient cells of columns x0 .. x0 + sw (the right tap of the last owned column)
//   sit in LDS for the whole item -- and processes exactly the pairs whose LEFT tap column X = floor(j rx + b_n) lies in
//   the slab.  With the keys of a problem sorted by b, the keys that meet that for BEV column j are one contiguous run
//   [kbeg, kend) of the sorted list, sliding by rx keys' worth per column (bevr_attn_slab_prep finds the runs).
//
// What that buys over the query-tile kernel (attn_bwd_q.hip: a (31 x 8)-query tile with a table window that MOVES with
// the keys, 0.33-0.48 moves per 64-key step, 93 GB of flush atomics per launch at the benchmark):
//   * no window logic at all: no region moves, refills, dirty boxes, grouped passes or global-memory path; any key set
//     runs at the same speed;
//   * every table cell leaves the workgroup ONCE per (item): the flush is ~(sw + 1) x rows floats per item, < 1 GB per
//     launch; what goes to memory instead is the slab's share of dQ, one float atomic per (query, channel) and
//     (slab, column) -- 2 S S 128 B x n_slab per (problem, head);
//   * 7 worker waves, one per row block of 31 queries, each running BOTH 32-key halves of a 64-key emission as two
//     interleaved dependency chains (8 waves per CU = 256 registers per wave: no spills; the first version, 14 workers of
//     one half each at 128 registers, reloaded spilled fragments from scratch on every emission and needed an exchange
//     to sum the two halves' dQ), and the staging, the per-(column, key) constants and the key bookkeeping on a PRODUCER
//     wave, one emission ahead: one barrier per emission, no staging registers in the workers.
// The price: K and V are re-streamed per slab (every pair belongs to one slab, every key to ~S sw / (Wt / rx) of them),
// and a workgroup is the whole CU (~150 KB of LDS).
//
// Work distribution: bevr_attn_slab_prep lists the non-empty (problem-head, slab, column chunk) items; the kernel is
// PERSISTENT, one workgroup per CU, items handed out by an atomic counter (every wave reaches the exit: the counter only
// grows and the list is finite).
//
// Rows.  Window row w of a slab column holds the pair (T2[w - ROW0], T2[w - ROW0 + 1]) of that table column;
// R = ROW0 + S + PADR + 31 n_rb + 2 rows cover every key with floor(a) in [-ROW0, S - 1 + PADR] for every row block
// (ROW0 = 10, PADR = 8: the learned offsets move a key ~3 rows past the table's ends).  A 32-key half with a key outside
// that range takes the CLAMPED body: the row index is clamped into the window, whose first and last rows lie outside the
// real table (zero values, gradient discarded), exactly as the zero padding of the global table.
#include "attn_tile.h"
#include "attn_tap.h"

#ifdef BEVR_SPROF
// phase stamps (make SPROF=1 OUTDIR=../lib_sprof; tools/prof_phases_slab.py): clocks summed over the emissions of wave 0
// (a worker), wave 6 (the last worker) and the producer of every workgroup
__device__ unsigned long long bevr_prof_slab[48];
extern "C" int bevr_debug_prof_slab(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[48] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(bevr_prof_slab), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bevr_prof_slab), 48 * 8);
}
__device__ __forceinline__ unsigned long long sprof_now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define SPROF(var) const unsigned long long var = sprof_now()
#define SPROF_ADD(i, v) pacc[i] += (v)
#else
#define SPROF(var)
#define SPROF_ADD(i, v)
#endif

namespace {

constexpr int SLAB_W_MAX = 24;     // owned table columns per slab (fewer when the rows of a large S leave less room)
constexpr int SLAB_ROW0 = 10;      // window row of table row 0
constexpr int SLAB_PADR = 8;       // rows past the table's last key row covered without clamping
constexpr int SLAB_RP = 448;       // row PITCH of a slab column in entries: 7 x 64, so that the two taps of a pair are ONE
                                   // ds_read2st64_b32 and the two gradient adds share an address register (S <= 211)
constexpr int SLAB_CH = 25;        // BEV columns per work item
constexpr int SQROWS = 31;         // query rows per row block (lane 31 of a half: the 32nd table row, attn_bwd_q.hip)
constexpr int SNRB = 7;            // row blocks per column (S <= 217)
constexpr int SNWORK = SNRB;       // worker waves: one per row block, BOTH 32-key halves of an emission each
constexpr int STHREADS = (SNWORK + 1) * 64;   // 8 waves = 2 per SIMD: 256 registers per wave (see the kernel)
constexpr int SEK = 64;            // keys per emission
constexpr int SKROW = 80;          // bytes per staged K / V row (64 + 16: conflict-free fragment reads)
constexpr float SLAB_EPS = 0.02f;  // slack of the key runs (in table columns): the kernel's own floor() decides membership

// per key, in the problem's b-sorted order (bevr_attn_slab_prep)
struct SlabKey { int A; float fy; float b; int pad; };

// per-(column, key) constants of an emission, written by the producer (lane = key), read as a broadcast
struct SlabCK {
  unsigned wA, wB;   // tap weights of column X / X + 1 as packed 16-bit pairs (row y, row y + 1)
  int cell;          // window index of the key's first tap for window row offset 0: colbase + row
  int row;           // its row part, A + ROW0 (the clamped body separates the two)
};

struct SlabLds {
  static constexpr int K_BYTES = SEK * SKROW;
  static constexpr int OFF_V = K_BYTES;
  static constexpr int OFF_CK = 2 * K_BYTES;
  static constexpr int OFF_CT = OFF_CK + SEK * 16;     // 2 x u32x4: {flags, j, amin0, amax0}, {amin1, amax1, 0, 0}
  static constexpr int BUF = OFF_CT + 32;
};
enum { SF_DONE = 1, SF_FIRST = 2, SF_LAST = 4 };

__host__ __device__ inline int slab_n_rb(int S) { return (S + SQROWS - 1) / SQROWS; }
__host__ __device__ inline int slab_rows(int S) { return SLAB_ROW0 + S + SLAB_PADR + SQROWS * slab_n_rb(S) + 2; }
// columns in LDS: column 0 is the KILL column (values -big, masked keys point there; its right neighbour is a real column,
// read with weight 0), columns 1 .. sw + 1 the slab's sw owned columns and the right tap column of the last one
__host__ __device__ inline size_t slab_lds_bytes(int S, int sw) {
  return (size_t)(sw + 2) * SLAB_RP * 12 + 2 * SlabLds::BUF + 64;
}
__host__ __device__ inline int slab_width(int S) {
  int sw = SLAB_W_MAX;
  while (sw > 1 && slab_lds_bytes(S, sw) > 160 * 1024) --sw;
  return sw;
}
// first table column any slab has to cover and the number of slabs: X = floor(j rx + b), b clamped as the key prep does
__host__ __device__ inline int slab_xmin(const bevr_attn_desc& d) { return -(d.Wt / 2 + 2) - 1; }
__host__ __device__ inline int slab_count(const bevr_attn_desc& d, int sw) {
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int xmax = (int)((float)(d.S - 1) * rx) + d.Wt + 3;
  return (xmax - slab_xmin(d) + sw) / sw;
}
__host__ __device__ inline int slab_chunks(int S) { return (S + SLAB_CH - 1) / SLAB_CH; }

// workspace: SlabKey[PG][N] | kbeg[PG][n_slab][S] | kend[PG][n_slab][S] | items[n_ph * n_slab * n_chunk] (int4) | counters
struct SlabWs {
  size_t off_beg, off_end, off_items, off_cnt, total;
  int n_slab, n_chunk, sw;
};
__host__ __device__ inline SlabWs slab_ws(const bevr_attn_desc& d) {
  SlabWs w;
  w.sw = slab_width(d.S);
  w.n_slab = slab_count(d, w.sw);
  w.n_chunk = slab_chunks(d.S);
  const size_t pg = (size_t)d.n_prob * d.groups;
  size_t o = pg * d.N * sizeof(SlabKey);
  o = (o + 255) & ~(size_t)255;
  w.off_beg = o;
  o += pg * w.n_slab * d.S * 4;
  w.off_end = o;
  o += pg * w.n_slab * d.S * 4;
  o = (o + 255) & ~(size_t)255;
  w.off_items = o;
  o += (size_t)d.n_prob * d.heads * w.n_slab * w.n_chunk * 16;
  w.off_cnt = o;
  o += 256;
  w.total = o;
  return w;
}

// ---------------------------------------------------------------------------------------------------------------
// preparation
__global__ __launch_bounds__(256) void slab_keys_kernel(bevr_attn_desc d, const float* __restrict__ key_a,
                                                        const float* __restrict__ key_b, const int* __restrict__ order,
                                                        SlabKey* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)d.n_prob * d.groups * d.N;
  if (t >= total) return;
  const int pg = (int)(t / d.N);
  const int n = order[t];
  float a = key_a[(size_t)pg * d.Np + n], b = key_b[(size_t)pg * d.Np + n];
  // the clamps of attn_keyprep.hip: every tap of a clamped key lies inside the zero-padded table
  const float aL = -(float)(d.Sp + 1), aU = (float)(d.Ht + 1);
  const float half = (float)(d.Wt / 2);
  const float bL = -(half + 2.0f), bU = (float)(d.Wt + 1);
  a = fminf(fmaxf(a, aL), aU);
  b = fminf(fmaxf(b, bL), bU);
  const float af = floorf(a);
  SlabKey k;
  k.A = (int)af;
  k.fy = a - af;
  k.b = b;
  k.pad = 0;
  out[t] = k;
}

// first index in the sorted keys of `pg` whose b is >= thr
__device__ __forceinline__ int slab_lower_bound(const SlabKey* __restrict__ keys, int N, float thr) {
  int lo = 0, hi = N;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid].b < thr) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// one thread per (pg, slab, chunk): the key runs of the chunk's columns, and the chunk's work items (one per head of the
// group) if any run is non-empty
__global__ __launch_bounds__(256) void slab_ranges_kernel(bevr_attn_desc d, const SlabKey* __restrict__ keys,
                                                          int* __restrict__ kbeg, int* __restrict__ kend,
                                                          int4* __restrict__ items, int* __restrict__ cnt, int n_slab,
                                                          int n_chunk, int sw) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int n_pg = d.n_prob * d.groups;
  if (t >= n_pg * n_slab * n_chunk) return;
  // chunk fastest: the items of one slab are neighbours in the list (their K / V runs overlap: L2 reuse)
  const int chunk = t % n_chunk, slab = (t / n_chunk) % n_slab, pg = t / (n_chunk * n_slab);
  const SlabKey* kp = keys + (size_t)pg * d.N;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int x0 = slab_xmin(d) + slab * sw;
  int work = 0;
  const int j1 = min(d.S, (chunk + 1) * SLAB_CH);
  for (int j = chunk * SLAB_CH; j < j1; ++j) {
    const float jrx = (float)j * rx;
    const int b0 = slab_lower_bound(kp, d.N, (float)x0 - jrx - SLAB_EPS);
    const int b1 = slab_lower_bound(kp, d.N, (float)(x0 + sw) - jrx + SLAB_EPS);
    kbeg[((size_t)pg * n_slab + slab) * d.S + j] = b0;
    kend[((size_t)pg * n_slab + slab) * d.S + j] = b1;
    work += b1 - b0;
  }
  if (work > 0) {
    const int hpg = d.heads / d.groups;
    const int prob = pg / d.groups, grp = pg % d.groups;
    const int at = atomicAdd(cnt, hpg);
    for (int h = 0; h < hpg; ++h) items[at + h] = make_int4(prob * d.heads + grp * hpg + h, slab, chunk, work);
  }
}

// the value of the lane below (lane - 1) across the whole wave; lane 0 receives 0 (v_mov_b32_dpp wave_shr:1)
__device__ __forceinline__ float slab_lane_below(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ unsigned long long slab_from_int(int v) {
  return ((unsigned long long)(unsigned)(v >> 31) << 32) | (unsigned)v;
}
__device__ __forceinline__ float slab_to_float(unsigned long long v) {
  const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  if (hi == (lo >> 31)) return (float)lo;
  return (float)(long long)v;
}
__device__ __forceinline__ int half_max_i(int v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v = max(v, __shfl_xor(v, s));
  return v;
}

// LDS-only barrier is not enough here: the producer's global loads are consumed by its own LDS stores before the barrier
#define SLAB_BARRIER() __syncthreads()

template <int PREC>
__global__ __launch_bounds__(STHREADS, 1) void attn_slab_bwd_q_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ Ks, const char* __restrict__ Vs,
    const SlabKey* __restrict__ skeys, const int* __restrict__ kbeg, const int* __restrict__ kend,
    const int4* __restrict__ items, int* __restrict__ cnt, const char* __restrict__ table_pair,
    const char* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ delta,
    const float* __restrict__ grad_scale, float* __restrict__ dQ, float* __restrict__ dtable, int n_slab, int sw, int R) {
  typedef SlabLds L;
  extern __shared__ __attribute__((aligned(256))) char lds[];
  // layout: vals (u32) | cells (u64) | staging x 2 | item slot
  constexpr int RP = SLAB_RP;
  const int ncell = (sw + 2) * RP;
  unsigned* vals = reinterpret_cast<unsigned*>(lds);
  unsigned long long* cells = reinterpret_cast<unsigned long long*>(lds + (size_t)ncell * 4);
  char* stage = reinterpret_cast<char*>(cells) + (size_t)ncell * 8;
  int* item_slot = reinterpret_cast<int*>(stage + 2 * L::BUF);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const bool producer = wave == SNWORK;
  const int rb = wave;                               // worker: row block
  const int n_rb = slab_n_rb(d.S);
  const int Mp = d.S * d.Sp;
  const int Hq = d.Hp + 1;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const int n_items = cnt[0];
  const int hpg = d.heads / d.groups;

  const float gscale = grad_scale[0], ginv = grad_scale[1] * BEVR_LN2;
  const float kp16 = PREC == BEVR_PREC_F16 ? grad_scale[2] : 0.f, c2_16 = PREC == BEVR_PREC_F16 ? grad_scale[3] : 1.f;
  const float cfix = PREC == BEVR_PREC_F16 ? grad_scale[0] * grad_scale[4] : 1.f;
  const float dq_scale = PREC == BEVR_PREC_F16 ? grad_scale[4] * BEVR_LN2 : ginv;
#ifdef BEVR_SPROF
  unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

  for (;;) {
    // ---- next work item -------------------------------------------------------------------------------------------
    SLAB_BARRIER();                                   // everybody is done with the previous item's slot and slab
    if (tid == 0) item_slot[0] = atomicAdd(cnt + 1, 1);
    SLAB_BARRIER();
    // read through readfirstlane: an LDS load is divergent to the compiler, and everything derived from the item (the
    // problem's pointers, the slab's origin, the chunk's columns) would live in vector registers
    const int it = __builtin_amdgcn_readfirstlane(item_slot[0]);
    if (it >= n_items) break;                         // uniform: every wave leaves here
    const int4 item = items[it];
    const int ph = __builtin_amdgcn_readfirstlane(item.x), slab = __builtin_amdgcn_readfirstlane(item.y),
              chunk = __builtin_amdgcn_readfirstlane(item.z);
    const int prob = ph / d.heads, hd = ph % d.heads;
    const int pg = prob * d.groups + hd / hpg;
    const int qb = prob / d.q_div;
    const int x0 = slab_xmin(d) + slab * sw;
    const int j0 = chunk * SLAB_CH, j1 = min(d.S, j0 + SLAB_CH);
    const char* tbl = table_pair + (size_t)hd * d.Wp * d.Hp * 8;
    float* dtb = dtable + (size_t)hd * d.Wp * Hq;
    SPROF(ti0);

    // ---- the slab: values in, cells cleared -----------------------------------------------------------------------
    for (int u = tid; u < (sw + 1) * RP; u += STHREADS) {
      const int c = u / RP, w = u - c * RP;
      const int xc = x0 + c + d.x_off;                // padded table column
      const int yr = w - SLAB_ROW0 + d.y_off;         // padded table row
      f32x2 v = {0.f, 0.f};
      if (w < R && yr >= 0 && yr < d.Hp && xc >= 0 && xc < d.Wp)   // tiny S: the window is taller than the padded table
        v = *reinterpret_cast<const f32x2*>(tbl + ((size_t)xc * d.Hp + yr) * 8);
      vals[RP + u] = Half<PREC>::pack2(v[0], v[1]);
      cells[RP + u] = 0ull;
    }
    for (int u = tid; u < RP; u += STHREADS) {
      vals[u] = Half<PREC>::pack2(Half<PREC>::NEG_BIG, Half<PREC>::NEG_BIG);
      cells[u] = 0ull;
    }
    SLAB_BARRIER();
    SPROF(ti1);
    SPROF_ADD(12, ti1 - ti0);     // slab in
    SPROF_ADD(15, 1);             // items

    if (producer) {
      // =========================================== PRODUCER ========================================================
      __builtin_amdgcn_s_setprio(3);
      const SlabKey* kp = skeys + (size_t)pg * d.N;
      const char* Kh = Ks + (size_t)ph * d.N * 64;
      const char* Vh = Vs + (size_t)ph * d.N * 64;
      // the chunk's key runs: lane c holds column j0 + c
      const size_t rbase = ((size_t)pg * n_slab + slab) * d.S;
      const int my_beg = j0 + lane < j1 ? kbeg[rbase + j0 + lane] : 0;
      const int my_end = j0 + lane < j1 ? kend[rbase + j0 + lane] : 0;
      // emission iterator: (column index c, first key k0) of the next emission to LOAD
      int lc = 0, lk = 0, lend = 0;
      auto seek = [&]() {       // from column index lc on: the first column with a non-empty run; lc == j1 - j0: none left
        while (lc < j1 - j0) {
          const int b = __builtin_amdgcn_readlane(my_beg, lc), e = __builtin_amdgcn_readlane(my_end, lc);
          if (b < e) { lk = b; lend = e; return; }
          ++lc;
        }
      };
      seek();
      u32x4 kv[8];
      SlabKey sk;
      int cur_c = -1, cur_k = 0, cur_end = 0;
      bool have = false;
      auto issue = [&]() {      // loads of the emission at (lc, lk); then step the iterator
        have = lc < j1 - j0;
        if (!have) return;
        cur_c = lc; cur_k = lk; cur_end = lend;
        const int kmax = d.N - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cid = lane + 64 * q;                       // chunk of the K tile: key cid / 4, part cid % 4
          const int key = min(cur_k + (cid >> 2), kmax);
          kv[q] = *reinterpret_cast<const u32x4*>(Kh + (size_t)key * 64 + (cid & 3) * 16);
          kv[4 + q] = *reinterpret_cast<const u32x4*>(Vh + (size_t)key * 64 + (cid & 3) * 16);
        }
        sk = kp[min(cur_k + lane, kmax)];
        lk += SEK;
        if (lk >= lend) { ++lc; seek(); }
      };
      issue();
      int e = 0;
      int prev_c = -1;
      while (have) {
        char* bb = stage + (e & 1) * L::BUF;
        const int c = cur_c, k0 = cur_k, kend_c = cur_end;
        const int j = j0 + c;
        const float jrx = (float)j * rx;
        // is this the column's last emission?  (the iterator already points at the next one)
        const bool last = !(lc < j1 - j0) || lc != c;
        const bool first = c != prev_c;
        prev_c = c;
        SPROF(tp0);
        const SlabKey sk_e = sk;
        u32x4 kv_e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kv_e[q] = kv[q];
        SPROF(tp1);
        SPROF_ADD(0, tp1 - tp0);      // producer: wait for this emission's loads
        issue();              // the next emission's loads fly while this one is written and processed
        // ---- constants of (column j, key) ----
        {
          const float tx = jrx + sk_e.b;
          const float xf = floorf(tx);
          const int X = (int)xf;
          const bool live = (k0 + lane < kend_c) && X >= x0 && X < x0 + sw;
          const float fx = tx - xf, fy = sk_e.fy;
          SlabCK ck;
          if (live) {
            ck.wA = Half<PREC>::pack2((1.0f - fx) * (1.0f - fy), (1.0f - fx) * fy);
            ck.wB = Half<PREC>::pack2(fx * (1.0f - fy), fx * fy);
            ck.row = sk_e.A + SLAB_ROW0;
            ck.cell = ((X - x0 + 1) * RP + ck.row) * 4;       // BYTE offset of the first tap's value entry
          } else {          // masked: first tap in the kill column => P = 0, dS = 0
            ck.wA = Half<PREC>::pack2(1.f, 0.f);
            ck.wB = 0u;
            ck.row = 0;
            ck.cell = 0;
          }
          *reinterpret_cast<SlabCK*>(bb + L::OFF_CK + lane * 16) = ck;
          // per half: any live key; any live key whose rows leave the window for some row block (the clamped body)
          const bool far = live && (sk_e.A + SLAB_ROW0 < 0 || sk_e.A > d.S - 1 + SLAB_PADR);
          const unsigned long long lm = __ballot(live), fm = __ballot(far);
          const unsigned hb = ((unsigned)lm != 0u ? 1u : 0u) | ((unsigned)(lm >> 32) != 0u ? 2u : 0u) |
                              ((unsigned)fm != 0u ? 4u : 0u) | ((unsigned)(fm >> 32) != 0u ? 8u : 0u);
          // the column after this one (the iterator is one emission ahead): the workers prefetch its Q / dO rows
          const int jn = (last && have) ? j0 + cur_c + 1 : 0;
          if (lane == 0)
            *reinterpret_cast<u32x4*>(bb + L::OFF_CT) =
                u32x4{(unsigned)((first ? SF_FIRST : 0) | (last ? SF_LAST : 0)) | (hb << 8), (unsigned)j | ((unsigned)jn << 16), 0u, 0u};
        }
        // ---- K and V rows ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cid = lane + 64 * q;
          *reinterpret_cast<u32x4*>(bb + (cid >> 2) * SKROW + (cid & 3) * 16) = kv_e[q];
          *reinterpret_cast<u32x4*>(bb + L::OFF_V + (cid >> 2) * SKROW + (cid & 3) * 16) = kv_e[4 + q];
        }
        ++e;
        SPROF(tp2);
        SPROF_ADD(1, tp2 - tp1);      // producer: constants, stores, next loads issued
        SLAB_BARRIER();
        SPROF(tp3);
        SPROF_ADD(2, tp3 - tp2);      // producer: waiting for the workers
        SPROF_ADD(3, 1);
      }
      if (lane == 0) *reinterpret_cast<u32x4*>(stage + (e & 1) * L::BUF + L::OFF_CT) = u32x4{(unsigned)SF_DONE, 0u, 0u, 0u};
      SLAB_BARRIER();
      __builtin_amdgcn_s_setprio(0);
    } else {
      // =========================================== WORKERS =========================================================
      const bool active = rb < n_rb;
      // waves 4 .. 6 share a SIMD with waves 0 .. 2, and the issue arbitration favours the older wave: the phase stamps had
      // wave 6's key-row loop 26 % longer than wave 0's, and the barrier waits for the slowest
      if (wave >= 4) __builtin_amdgcn_s_setprio(1);
      const int i0 = rb * SQROWS;
      const int qrow = i0 + lq;
      const bool live = lq < SQROWS && qrow < d.S;
      const int rowoff4 = (i0 + lq) * 4;
      const char* Qh = Q + ((size_t)(qb * d.heads + hd) * Mp) * 64;
      const char* dOh = dO + ((size_t)ph * Mp) * 64;
      const unsigned cells_off = (unsigned)(reinterpret_cast<char*>(cells) - lds);
      Frag<PREC> qf, dof;
      float nl = 0.f, nd = 0.f, lse_r = 0.f, dlt_r = 0.f;
      int jcur = -1, jpend = -1;
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;

      // the rows of column j for this wave's row block: issued (raw) ...
      auto issue_column = [&](int j) {
        const size_t mq = (size_t)j * d.Sp + min(qrow, d.S - 1);
        qf.load(Qh + mq * 64, hi);
        dof.load(dOh + mq * 64, hi);
        lse_r = LSE[(size_t)ph * Mp + mq];
        dlt_r = delta[(size_t)ph * Mp + mq];
        jpend = j;
      };
      // ... and made ready: lanes without a query compute on a copy of a real one with dO = delta = 0 (their dS is exactly
      // 0); the fixed-point scale of the table-gradient cells is folded into dO and delta (attn_bwd_q.hip)
      auto finish_column = [&]() {
        float dlt = dlt_r;
        if (!live) {
          dof.v[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          dof.v[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
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
        }
        nl = kp16 - lse_r;
        nd = -dlt;
        jcur = jpend;
      };
      // one flush per (slab, column): this wave saw every key of the column's run
      auto flush_dq = [&]() {
        // dq[r]: query i0 + crow(r, hi), channel lq (dq_product): a wave instruction adds two 128-byte rows
        float* base = dQ + ((size_t)ph * Mp + (size_t)jcur * d.Sp + i0) * 32 + lq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qr = crow(r, hi);
          if (qr < SQROWS && i0 + qr < d.S) atomicAdd(base + (size_t)qr * 32, dq_scale * dq[r]);
          dq[r] = 0.f;
        }
      };

      // One emission -- both 32-key halves, one after the other -- against this wave's 31 queries.  The key-row loop of a
      // half runs TWO key rows per step (rows r and r + 8: two independent dependency chains written side by side): with
      // two waves per SIMD the latency of a chain (LDS round trips: constants -> taps -> ... -> atomics) is covered by
      // the other chain and the other wave, where the 14-wave version (one half per wave, one row per step) needed
      // 128-register waves that spilled.  The matrix products are placed so that they run UNDER a loop: S and dP of the
      // second half are issued before the first half's loop, the first half's dQ product before the second half's loop.
      // CLAMP: the row index is clamped into the window.  jn >= 0: the column's last emission -- once the products that
      // read Q and dO are issued, the rows of column jn are requested INTO the same registers.
      auto process = [&](const char* bb, int jn, auto clamp_tag) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        const SlabCK* pk0 = reinterpret_cast<const SlabCK*>(bb + L::OFF_CK);
        f32x16 s[2], dp[2];
        {
          float a = nl, b = nd;
          asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[h][r] = a; dp[h][r] = b; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          Frag<PREC> kf, vkf;
          kf.load(bb + (h * 32 + lq) * SKROW, hi);
          s[h] = mma_frag(kf, qf, s[h]);          // S^T - LSE
          vkf.load(bb + L::OFF_V + (h * 32 + lq) * SKROW, hi);
          dp[h] = mma_frag(vkf, dof, dp[h]);      // dP^T - delta
        }
        if (jn >= 0) issue_column(jn);
        __builtin_amdgcn_sched_barrier(0);
        SPROF(tq0);
        // byte offset of the key's first tap VALUE for this lane's row; the gradient cell sits at twice that (8-byte
        // cells behind the 4-byte values)
        auto offset = [&](const SlabCK& e) -> int {
          if constexpr (CLAMP) return (e.cell - 4 * e.row) + 4 * max(0, min(e.row + (rowoff4 >> 2), R - 1));
          else return e.cell + rowoff4;
        };
        auto read_tap = [&](int off, unsigned& a, unsigned& b) {
          a = *reinterpret_cast<const unsigned*>(lds + off);
          b = *reinterpret_cast<const unsigned*>(lds + off + RP * 4);
        };
        auto dq_product = [&](int h) {
          // A operand K^T[channel lq][key] for the accumulator contraction: element j of k-step s <-> key
          // 16 s + 8 (j >> 2) + 4 hi + (j & 3) (bevr_common.h: mma_acc_b), out of the row tile by transposed reads
          Frag<PREC> ktf;
          const int i16 = lane & 15, chalf = (lane >> 4) & 1;
          const char* p = bb + (h * 32 + 4 * hi + (i16 >> 2)) * SKROW + chalf * 32 + 8 * (i16 & 3);
          ktf.v[0] = lds_tr8(p, 8 * SKROW);
          ktf.v[1] = lds_tr8(p + 16 * SKROW, 8 * SKROW);
          // dQ[query][channel] += dS[query][key] K[key][channel]: the accumulator tile of dS^T (key rows in registers, query
          // on the lane) is the A operand as it stands -- lane (query, hi) holds the keys crow(8 s + j, hi) of k-step s,
          // the same key order as ktf's -- and the product comes out with the CHANNEL on the lane: the flush below adds
          // 128 contiguous bytes per query row (with the query on the lane, as the first version had it, every lane of an
          // atomic instruction hit its own cache line: 85 GB of atomic write traffic per launch against ~23 GB of payload)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            u32x4 w;
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = Half<PREC>::pack2(s[h][8 * ks + 2 * k], s[h][8 * ks + 2 * k + 1]);
            dq = Half<PREC>::mfma(__builtin_bit_cast(bf16x8, w), ktf.v[ks], dq);
          }
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const SlabCK* pk = pk0 + h * 32;
          // chain c walks the key rows c * 8 + 0 .. 7 of the accumulator tile
          SlabCK e0[2], e1[2];
          int o0[2];
          unsigned ta[2], tb[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            e0[c] = pk[crow(8 * c, hi)];
            e1[c] = pk[crow(8 * c + 1, hi)];
            o0[c] = offset(e0[c]);
            read_tap(o0[c], ta[c], tb[c]);
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            unsigned na[2], nb[2];
            SlabCK e2[2];
            int o1[2];
            // the LDS atomics are ordered memory operations for the compiler (it moves no load across them): the taps of
            // the next key row and the constants of the one after, of BOTH chains, are requested before the adds of this
            // step are issued
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              na[c] = ta[c]; nb[c] = tb[c]; e2[c] = e1[c]; o1[c] = o0[c];
              if (r + 1 < 8) { o1[c] = offset(e1[c]); read_tap(o1[c], na[c], nb[c]); }
              if (r + 2 < 8) e2[c] = pk[crow(8 * c + r + 2, hi)];
            }
            int iA[2], iB[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int row = 8 * c + r;
              float sv = Half<PREC>::dot2(ta[c], e0[c].wA, s[h][row]);
              sv = Half<PREC>::dot2(tb[c], e0[c].wB, sv);
              float ds = fast_exp2(sv) * dp[h][row];
              if constexpr (PREC == BEVR_PREC_F16) ds *= c2_16;
              s[h][row] = ds;
              const float gb_ = slab_lane_below(ds);
              if constexpr (PREC == BEVR_PREC_BF16) {
                const unsigned pr = pack_bf16x2(ds, gb_);
                asm("v_dot2_f32_bf16 %0, %2, %3, 0\n\t"
                    "v_dot2_f32_bf16 %1, %2, %4, 0\n\t"
                    "s_nop 2\n\t"
                    "v_cvt_rpi_i32_f32 %0, %0\n\t"
                    "v_cvt_rpi_i32_f32 %1, %1"
                    : "=&v"(iA[c]), "=&v"(iB[c])
                    : "v"(pr), "v"(e0[c].wA), "v"(e0[c].wB));
              } else {
                const unsigned pr = Half<PREC>::pack2(ds, gb_);
                asm("v_dot2_f32_f16 %0, %2, %3, 0\n\t"
                    "v_dot2_f32_f16 %1, %2, %4, 0\n\t"
                    "s_nop 2\n\t"
                    "v_mul_f32 %0, %0, %5\n\t"
                    "v_mul_f32 %1, %1, %5\n\t"
                    "v_cvt_rpi_i32_f32 %0, %0\n\t"
                    "v_cvt_rpi_i32_f32 %1, %1"
                    : "=&v"(iA[c]), "=&v"(iB[c])
                    : "v"(pr), "v"(e0[c].wA), "v"(e0[c].wB), "v"(cfix));
              }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              unsigned long long* gp = reinterpret_cast<unsigned long long*>(lds + cells_off + 2 * o0[c]);
              atomicAdd(gp, slab_from_int(iA[c]));
              atomicAdd(gp + RP, slab_from_int(iB[c]));
              e0[c] = e1[c]; e1[c] = e2[c]; ta[c] = na[c]; tb[c] = nb[c]; o0[c] = o1[c];
            }
          }
          // the half's dQ product: issued here, it runs under the other half's loop (h = 0) or the barrier wait (h = 1)
          dq_product(h);
          __builtin_amdgcn_sched_barrier(0);
        }
        SPROF(tq1);
        SPROF_ADD(4, tq1 - tq0);      // worker: the key-row loops and dQ products
      };

      int e = 0;
      for (;;) {
        SPROF(tw0);
        SLAB_BARRIER();
        SPROF(tw1);
        SPROF_ADD(0, tw1 - tw0);      // worker: barrier wait
        const char* bb = stage + (e & 1) * L::BUF;
        const u32x4 ct = *reinterpret_cast<const u32x4*>(bb + L::OFF_CT);
        const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)ct[0]);
        const unsigned jword = (unsigned)__builtin_amdgcn_readfirstlane((int)ct[1]);
        if (flags & SF_DONE) break;
        if (active) {
          const int j = (int)(jword & 0xffffu);
          if (flags & SF_FIRST) {
            if (jpend != j) issue_column(j);      // not prefetched: an item's first column
            finish_column();
          }
          const unsigned hb = flags >> 8;          // bits 0, 1: the halves with a live key; bits 2, 3: with a far one
          if (hb & 3u) {
            const int jn = (flags & SF_LAST) ? (int)(jword >> 16) - 1 : -1;
            if (hb & 12u) process(bb, jn, std::true_type{});
            else process(bb, jn, std::false_type{});
          }
          if (flags & SF_LAST) flush_dq();
        }
        SPROF(tw2);
        SPROF_ADD(1, tw2 - tw1);      // worker: the emission
        SPROF_ADD(2, (flags >> 8) & 1u);
        SPROF_ADD(3, 1);
        ++e;
      }
    }

    // ---- flush the slab: every cell once ----------------------------------------------------------------------------
    SLAB_BARRIER();
    SPROF(tf0);
    for (int u = tid; u < (sw + 1) * RP; u += STHREADS) {
      const int c = u / RP, w = u - c * RP;
      const unsigned long long v = w < R ? cells[RP + u] : 0ull;
      if (v != 0ull) {
        const int xc = x0 + c + d.x_off, yr = w - SLAB_ROW0 + d.y_off;
        if (xc >= 0 && xc < d.Wp && yr >= 0 && yr < Hq) atomicAdd(dtb + (size_t)xc * Hq + yr, slab_to_float(v) * ginv);
      }
    }
    SPROF(tf1);
    SPROF_ADD(13, tf1 - tf0);     // slab out
  }
#ifdef BEVR_SPROF
  if (lane == 0 && (wave == 0 || wave == SNWORK - 1 || producer)) {
    const int base = producer ? 32 : (wave ? 16 : 0);
    for (int i = 0; i < 16; ++i) atomicAdd(&bevr_prof_slab[base + i], pacc[i]);
  }
#endif
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* Ks, const void* Vs, const void* ws, const float* table_pair,
           const void* dO, const float* LSE, const float* delta, const float* grad_scale, float* dQ, float* dtable,
           hipStream_t st) {
  const SlabWs w = slab_ws(d);
  const char* base = static_cast<const char*>(ws);
  int* cnt = reinterpret_cast<int*>(const_cast<char*>(base) + w.off_cnt);
  hipError_t e = hipMemsetAsync(cnt + 1, 0, 4, st);       // the item counter of this launch
  if (e != hipSuccess) return (int)e;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return BEVR_E_SHAPE;
    n_cu = prop.multiProcessorCount;
  }
  const int R = slab_rows(d.S);
  const size_t lds = slab_lds_bytes(d.S, w.sw);
  static bool attr_set[4] = {false, false, false, false};
  if (!attr_set[PREC]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_slab_bwd_q_kernel<PREC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set[PREC] = true;
  }
  hipLaunchKernelGGL((attn_slab_bwd_q_kernel<PREC>), dim3(n_cu), dim3(STHREADS), lds, st, d, (const char*)Q,
                     (const char*)Ks, (const char*)Vs, reinterpret_cast<const SlabKey*>(base),
                     reinterpret_cast<const int*>(base + w.off_beg), reinterpret_cast<const int*>(base + w.off_end),
                     reinterpret_cast<const int4*>(base + w.off_items), cnt, (const char*)table_pair, (const char*)dO,
                     LSE, delta, grad_scale, dQ, dtable, w.n_slab, w.sw, R);
  return (int)hipGetLastError();
}

bool slab_shape_ok(const bevr_attn_desc& d) {
  return is16(d.precision) && slab_n_rb(d.S) <= SNRB && slab_rows(d.S) <= SLAB_RP && SLAB_CH <= 64 &&
         slab_lds_bytes(d.S, slab_width(d.S)) <= 160 * 1024;
}

}  // namespace

extern "C" size_t bevr_attn_slab_ws_bytes(const bevr_attn_desc* d) {
  if (bevr_check_desc(d) || !slab_shape_ok(*d)) return 0;
  return slab_ws(*d).total;
}

extern "C" int bevr_attn_slab_prep(const bevr_attn_desc* d, const float* key_a, const float* key_b, const int* order,
                                   void* slab_ws_, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!key_a || !key_b || !order || !slab_ws_) return BEVR_E_NULL;
  if (!is16(d->precision)) return BEVR_E_PRECISION;
  if (!slab_shape_ok(*d)) return BEVR_E_SHAPE;
  if (!bevr_aligned16(slab_ws_)) return BEVR_E_ALIGN;
  const SlabWs w = slab_ws(*d);
  char* base = static_cast<char*>(slab_ws_);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(base + w.off_cnt, 0, 256, st);
  if (e != hipSuccess) return (int)e;
  const size_t nk = (size_t)d->n_prob * d->groups * d->N;
  hipLaunchKernelGGL(slab_keys_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, *d, key_a, key_b, order,
                     reinterpret_cast<SlabKey*>(base));
  const int nt = d->n_prob * d->groups * w.n_slab * w.n_chunk;
  hipLaunchKernelGGL(slab_ranges_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, *d,
                     reinterpret_cast<const SlabKey*>(base), reinterpret_cast<int*>(base + w.off_beg),
                     reinterpret_cast<int*>(base + w.off_end), reinterpret_cast<int4*>(base + w.off_items),
                     reinterpret_cast<int*>(base + w.off_cnt), w.n_slab, w.n_chunk, w.sw);
  return (int)hipGetLastError();
}

extern "C" int bevr_attn_slab_bwd_q(const bevr_attn_desc* d, const void* Q, const void* Ks, const void* Vs,
                                    const void* slab_ws_, const float* table_pair, const void* dO, const float* LSE,
                                    const float* delta, const float* grad_scale, float* dQ, float* dtable, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !Ks || !Vs || !slab_ws_ || !table_pair || !dO || !LSE || !delta || !grad_scale || !dQ || !dtable)
    return BEVR_E_NULL;
  if (!is16(d->precision)) return BEVR_E_PRECISION;
  if (!slab_shape_ok(*d)) return BEVR_E_SHAPE;
  if (!bevr_aligned16(Q) || !bevr_aligned16(Ks) || !bevr_aligned16(Vs) || !bevr_aligned16(dO) || !bevr_aligned16(dQ) ||
      !bevr_aligned16(table_pair) || !bevr_aligned16(slab_ws_))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16)
    return launch<BEVR_PREC_BF16>(*d, Q, Ks, Vs, slab_ws_, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st);
  return launch<BEVR_PREC_F16>(*d, Q, Ks, Vs, slab_ws_, table_pair, dO, LSE, delta, grad_scale, dQ, dtable, st);
}
