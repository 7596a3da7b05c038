// Forward of the GATHER kernels: attention over SCATTERED keys (the camera-visible keys of SCA in their static k-d
// order, TSA's keys) with every product on the matrix cores, 16-bit operand modes.
//     S[n][q] = scale Q_q . K_n + bias[n][q],   bias = bilinear sample of the rpe table at (i + a_n, j rx + b_n)
//     O_q     = softmax_n(S) V                  (model/SCA_deform_attn.py:331-413, model/TSA_deform_attn.py:245-333)
// Same operands, key workspace and outputs (O normalised, two LSE planes) as bevr_attn_fwd (attn_fwd.hip), which it
// replaces for the 16-bit modes; the table arrives as 16-bit PAIRS, table_pk[h][Wp][Hp] = (T2[y][x], T2[y + 1][x]).
//
// The bias as a matrix product with a SPARSE contraction: a 16-key tile has 64 slots, slot (key n', tap) -- the four
// bilinear taps of key n' in the order (y, x), (y + 1, x), (y, x + 1), (y + 1, x + 1):
//     bias^T[n][q] = sum_slots W[n][slot] Tg[slot][q],    W[n][(n', tap)] = w_tap(n) [n' == n],
//                                                          Tg[(n', tap)][q] = T2[A_n' + i_q + dy][X_n' + dx]
// Tg is the B operand exactly as the lanes GATHER it: lane (q, k-group) holds one dword (a (y, y + 1) pair) of each of four
// keys at ONE of their two table columns -- ds_read_b32 from an LDS WINDOW of the pair table; the 16 lanes of a k-group
// read 16 consecutive rows of one column, the two k-groups of a half wave the columns x and x + 1 of the same key, a
// column stride = 16 (mod 32 banks) apart: no bank conflicts wherever the key lies (round 4's slot order -- two keys x two
// columns per lane, the k-groups on DIFFERENT keys -- lost 34 % of the LDS cycles to them).  W is block diagonal: zero
// except for the 8 bytes (four packed weights) each key owns.  Per 16 keys x 16 BEV rows: 3 MFMAs for S (QK^T + 2 x bias), 1 for
// PV, 4 LDS gathers; no per-pair arithmetic but the exponential.
//
// Work split (as the tap kernels, attn_tap_fwd.hip): workgroup = ONE BEV column j of one (problem, head); wave w owns
// the 16-row blocks [w NB, (w + 1) NB) of the column; the LAST wave is the PRODUCER.  Per emission (one 32-key tile) it
// stages the K rows, the V rows (two 16-channel images for the transposed reads of the PV product), the W image, every
// key's window address, and the WINDOW itself: the columns [x0, x1] x rows [a0, a0 + ROWS) of the pair table that the
// tile's taps reach for ALL BEV rows of the column (a k-d leaf of 32 keys: ~13 columns x 256 rows, 13 KB, one 16-byte
// load per lane and column).  Two buffers: the producer fills emission e + 1 while the row-block waves work on e; one
// barrier per emission.
//
// ANY key set is handled: a tile whose box does not fit the window (more than WIN_COLS columns or ROWS rows) is emitted
// in groups of 14 keys with one two-column STRIP of the table per key, the other keys masked.
//
// Softmax reference: STATIC.  The first pass works against the caller's mref[q] (an upper bound of the row's logits minus
// a headroom, bevrender_amd/ops.py): no running maximum, rescale or mass test in the loop.  A row whose weights all
// underflowed against it flags its column and the EXACT instantiation -- the only one with an online maximum (the
// maximum of the row's first tile, raised with a rescale when a later logit exceeds it) -- recomputes that column.
// The row's largest weight is tracked beside the sums for LSE plane 1.
#include "attn_tap.h"

#ifdef BEVR_GPROF
__device__ unsigned long long bevr_prof_gather[32];
extern "C" int bevr_debug_prof_gather(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(bevr_prof_gather), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bevr_prof_gather), 32 * 8);
}
__device__ __forceinline__ unsigned long long gprof_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define GPROF(var) const unsigned long long var = gprof_now()
#define GPROF_ADD(i, v) gacc[i] += (v)
#else
#define GPROF(var)
#define GPROF_ADD(i, v)
#endif

namespace {

constexpr int GT = 32;                 // keys per emission (one 32-key tile; 64: two tiles that share one window -- too
                                       // wide for WIN_COLS at the benchmark: 1.7x the emissions)
constexpr int NT = GT / 32;
constexpr int ROWS = 256;              // window rows (dwords) per column: one 16-byte load per lane
constexpr int PITCH = 272;             // column stride in dwords, = 16 (mod 32): the two 16-lane k-groups of a half wave read
                                       // rows i .. i + 15 of columns x and x + 1 of ONE key -- opposite halves of the 32 banks
                                       // of ds_read_b32, conflict-free for every key position
constexpr int WIN_COLS = 28;
constexpr int STRIP_KEYS = WIN_COLS / 2;
struct LdsG {
  static constexpr int OFF_K = 0;                       // [64 keys][4 chunks of 16 B], chunk c of key n at position c ^ ((n >> 2) & 3)
  static constexpr int OFF_VLO = GT * 64;               // [64 keys][16 channels] 16-bit: channels 0..15
  static constexpr int OFF_VHI = OFF_VLO + GT * 32;     // channels 16..31
  static constexpr int OFF_W = OFF_VHI + GT * 32;       // the W operand AS THE LANES HOLD IT: [16-key sub-tile][bias product][lane]
                                                        // 16 bytes; zero but for one dword per (key, column): w_image_pos
  static constexpr int W_BYTES = (GT / 16) * 2 * 64 * 16;
  static constexpr int OFF_OFF = OFF_W + W_BYTES;       // [64 keys] LDS byte address of tap (y, x) for BEV row 0
  static constexpr int OFF_CT = OFF_OFF + GT * 4;       // u32x4: flags (bit 2: done), live keys 0..31, 32..63, entries of the fill list
  static constexpr int OFF_FD = OFF_CT + 16;            // fill list of the NEXT emission's window: [WIN_COLS] (column, first row)
  static constexpr int BUF = OFF_FD + WIN_COLS * 8;
  static constexpr int WIN = WIN_COLS * PITCH * 4;
  static constexpr int OFF_WIN = 2 * BUF;
  static constexpr int TOTAL = OFF_WIN + 2 * WIN;
};
static_assert(LdsG::BUF % 16 == 0 && LdsG::WIN % 16 == 0, "16-byte aligned LDS blocks");
static_assert(2 * LdsG::TOTAL <= 160 * 1024, "two workgroups per CU");
// Slot order of a bias product (8 keys x 4 taps = 32 slots of the contraction): k-group kg holds the dwords ((y, y + 1)
// pairs) of keys 4 (kg >> 1) + {0 .. 3} at table column x + (kg & 1).  The A operand of product m of a 16-key sub-tile is
// then, for lane (key li, k-group kg), dword dd = the key's packed weights of that column if li == 8 m + 4 (kg >> 1) + dd,
// else 0: byte offset in the W image of the dword that key n (0 .. 15) owns for column c (0, 1); the other column's is
// 256 bytes further.
__device__ __forceinline__ int w_image_pos(int n) {
  const int m = n >> 3, half = (n >> 2) & 1, dd = n & 3;
  return (m * 64 + n + 32 * half) * 16 + dd * 4;
}
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load in flight
// (s_waitcnt vmcnt(0)): the prefetches that are meant to cross the barrier.
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr float RAISE = 40.0f;   // EXACT: binades a logit may exceed the reference before the reference moves

template <int V> struct IntC { static constexpr int value = V; };
typedef const char __attribute__((address_space(1)))* gptr_t;
typedef char __attribute__((address_space(3)))* lptr_t;
// 64 lanes x 16 bytes, global (per-lane address) -> LDS (base + 16 lane), no registers
__device__ __forceinline__ void glds16(const char* src, char* lds_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_base, 16, 0, 0);
}

template <int PREC, int NB, bool EXACT, bool DMA>
__global__ __launch_bounds__(512, 4) void attn_gather_fwd_kernel(
    bevr_attn_desc d, const char* __restrict__ Q, const char* __restrict__ K, const char* __restrict__ V,
    const char* __restrict__ key_ws, const uint32_t* __restrict__ table_pk, const float* __restrict__ mref,
    float* __restrict__ O, float* __restrict__ LSE, int* __restrict__ flags) {
  typedef LdsG L;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int n_ph = d.n_prob * d.heads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ph = (slot / d.S) * 8 + xcd;
  if (ph >= n_ph) return;
  const int j = slot % d.S;
  if constexpr (EXACT) {
    if (flags[ph * d.S + j] == 0) return;
  }
  const int prob = ph / d.heads, hd = ph % d.heads;
  const int tid = threadIdx.x, n_wave = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, kg = lane >> 4;
  const int Mp = d.S * d.Sp;
  const int nblk = (d.S + QB - 1) / QB;
  const int rows_q = nblk * QB;
  const float rx = (float)(d.Wt - 1) / (2.0f * (float)(d.S - 1));
  const float jrx = (float)j * rx;
  const unsigned smem_base = (unsigned)(size_t)(lptr_t)smem;     // window addresses are handed over as LDS addresses

  // ---- the window fill, shared by ALL waves: entry c of a fill list = (padded table column, first table row) of window
  // column c; wave w copies the entries w, w + n_wave, ...  One column = ROWS dwords = 64 lanes x 16 bytes, global -> LDS
  // directly (no registers): issued when an emission starts, waited for before its closing barrier -- the L2 latency hides
  // behind the emission's matrix work.  A table shorter than a window column (small problems) goes through registers.
  const char* tbl = reinterpret_cast<const char*>(table_pk + (size_t)hd * d.Wp * d.Hp);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto fill_one = [&](char* win, int c, int xc, int r0) {      // xc, r0: uniform
    static_assert(ROWS == 256, "one 16-byte load per lane and column");
    if constexpr (DMA) {       // the launcher's choice: the table is at least a window column tall
      // scalar base + this lane's 16 bytes: no vector address arithmetic next to the accumulators
      const char* sbase = tbl + ((size_t)__builtin_amdgcn_readfirstlane(xc) * d.Hp + __builtin_amdgcn_readfirstlane(r0)) * 4;
      glds16(sbase + lane16, win + (c * PITCH) * 4);
    } else {
      const int rr = r0 + 4 * lane;
      const char* src = tbl + ((size_t)xc * d.Hp + rr) * 4;
      u32x4 v = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (rr + k < d.Hp) v[k] = *reinterpret_cast<const uint32_t*>(src + 4 * k);
      *reinterpret_cast<u32x4*>(win + (c * PITCH) * 4 + lane * 16) = v;
    }
  };
  auto fill_share = [&](const int* fd, char* win, int nfill) {
    for (int c = wave; c < nfill; c += n_wave) fill_one(win, c, fd[2 * c], fd[2 * c + 1]);
  };

  // the W images of both buffers: zero once; the producer rewrites the dwords the keys own every emission
  for (int i = tid; i < 2 * L::W_BYTES / 16; i += (int)blockDim.x)
    *reinterpret_cast<u32x4*>(smem + (i / (L::W_BYTES / 16)) * L::BUF + L::OFF_W + (i % (L::W_BYTES / 16)) * 16) =
        u32x4{0u, 0u, 0u, 0u};       // ordered before the first use by the prologue's barrier
  if (wave == n_wave - 1) {
    // ---- producer ------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    const int pg = prob * d.groups + hd / (d.heads / d.groups);
    const KeyW* kws = reinterpret_cast<const KeyW*>(key_ws) + (size_t)pg * d.Np;
    const StepBox* box = reinterpret_cast<const StepBox*>(key_ws + key_ws_box_offset(d)) + (size_t)pg * (d.Np / 32);
    const char* Kp = K + (size_t)ph * d.Np * 64;
    const char* Vp = V + (size_t)ph * d.Np * 64;
    const int n_steps = d.Np / GT;

    // One emission: the keys `sel` of a 64-key step, their weights and window addresses, and the fill list of its window
    // (this lane's entry).  The producer DESCRIBES one emission ahead of the one it stages (the fill list of window e + 1
    // travels with emission e) and stages one emission ahead of the row-block waves.
    struct Em {
      unsigned long long sel;
      int step, nfill, fd_xc, fd_r0, koff;
      u32x2 w;
    };
    int step = -1;
    unsigned long long rem = 0ull;
    const int kl = lane & (GT - 1);      // this lane's key of a step (GT = 32: the upper half wave duplicates the lower)
    const bool klane = lane < GT;
    KeyW kw_n = kws[kl];
    StepBox sb0_n = box[0], sb1_n = box[NT - 1];
    int ar = 0, xc = 0, x0 = 0, a0w = 0, cols = 0;
    bool fits = false;
    u32x2 w_t = {0u, 0u};
    auto advance = [&](Em& em) -> bool {
      while (rem == 0ull) {      // on to the next step with live keys
        if (++step >= n_steps) return false;
        const KeyW kw = kw_n;
        const StepBox sb = box_union(sb0_n, sb1_n);
        if (step + 1 < n_steps) {
          kw_n = kws[(size_t)(step + 1) * GT + kl];
          sb0_n = box[NT * (step + 1)];
          sb1_n = box[NT * (step + 1) + NT - 1];
        }
        if (sb.amax < sb.amin) continue;               // a step of padding only
        const bool live = klane && step * GT + kl < d.N;
        ar = (kw.aoff >> 3) - d.x_off * d.Hp;          // padded table row of tap (y, .) for BEV row 0
        const float tx = jrx + kw.b;
        const float xf = floorf(tx);
        xc = (int)xf + d.x_off;                        // padded table column of tap (., x)
        const float fx = tx - xf, fy = kw.fy;
        w_t[0] = Half<PREC>::pack2((1.0f - fx) * (1.0f - fy), (1.0f - fx) * fy);
        w_t[1] = Half<PREC>::pack2(fx * (1.0f - fy), fx * fy);
        // the step's box: columns [x0, x1], rows [a0, a0 + rspan)
        x0 = (int)floorf(jrx + sb.bmin) + d.x_off;
        const int x1 = (int)floorf(jrx + sb.bmax) + 1 + d.x_off;
        cols = x1 - x0 + 1;
        fits = cols <= WIN_COLS && sb.amax - sb.amin + rows_q + 1 <= ROWS;
        a0w = max(0, min(sb.amin + d.y_off, d.Hp - ROWS));
        rem = __ballot(live);
      }
      em.w = w_t;
      em.step = step;
      if (fits) {
        em.sel = rem;
        em.koff = ((xc - x0) * PITCH + (ar - a0w)) * 4;
        em.nfill = cols;
        em.fd_xc = x0 + lane;
        em.fd_r0 = a0w;
      } else {
        // the first STRIP_KEYS remaining keys, one two-column strip each: entries 2 rank + {0, 1}
        const int rank = __builtin_popcountll(rem & ((1ull << kl) - 1ull));
        const bool mine = klane && (rem >> kl & 1ull) && rank < STRIP_KEYS;
        em.sel = __ballot(mine);
        const int akw = max(0, min(ar, d.Hp - ROWS));
        em.koff = (2 * rank * PITCH + (ar - akw)) * 4;
        em.nfill = 2 * __builtin_popcountll(em.sel);
        // lane 2 r + c holds the entry of the key with rank r: fetch that key's (xc, akw)
        int src = 0;
        {
          unsigned long long t = em.sel;
          for (int k = 0; k < (lane >> 1) && t; ++k) t &= t - 1;
          src = t ? __builtin_ctzll(t) : 0;
        }
        em.fd_xc = __shfl(xc, src) + (lane & 1);
        em.fd_r0 = __shfl(akw, src);
      }
      rem &= ~em.sel;
      return true;
    };
    // Staging of emission e (one emission ahead of its use).  K / V rows go global -> LDS directly (8 x 1 KB, no
    // registers): K with its 16-byte chunks XOR-swizzled (conflict-free fragment reads of unpadded 64-byte rows), V as two
    // [key][16 channel] images.
    auto stage = [&](const Em& em, int e, const Em* nxt) {
      char* bb = smem + (e & 1) * L::BUF;
      const size_t n0 = (size_t)em.step * GT;
#pragma unroll
      for (int i = 0; i < GT / 16; ++i) {      // keys 16 i .. 16 i + 15: lane = (key, position)
        const int n = 16 * i + (lane >> 2), c = (lane & 3) ^ ((n >> 2) & 3);
        glds16(Kp + (n0 + n) * 64 + 16 * c, bb + L::OFF_K + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {           // keys 32 i .. 32 i + 31: lane = (key, 16-byte half of the 16 channels)
        const int n = 32 * i + (lane >> 1);
        glds16(Vp + (n0 + n) * 64 + 16 * (lane & 1), bb + L::OFF_VLO + i * 1024);
        glds16(Vp + (n0 + n) * 64 + 32 + 16 * (lane & 1), bb + L::OFF_VHI + i * 1024);
      }
      if (klane) {
        char* wk = bb + L::OFF_W + (kl >> 4) * 2048 + w_image_pos(kl & 15);
        *reinterpret_cast<uint32_t*>(wk) = em.w[0];
        *reinterpret_cast<uint32_t*>(wk + 256) = em.w[1];
        // a key outside the emission: any address inside the window (its logit is masked)
        const bool in = em.sel >> kl & 1ull;
        *reinterpret_cast<unsigned*>(bb + L::OFF_OFF + kl * 4) = smem_base + L::OFF_WIN + (e & 1) * L::WIN + (in ? em.koff : 0);
      }
      if (nxt && lane < nxt->nfill) {
        int* fd = reinterpret_cast<int*>(bb + L::OFF_FD);
        fd[2 * lane] = nxt->fd_xc;
        fd[2 * lane + 1] = nxt->fd_r0;
      }
      if (lane == 0)
        *reinterpret_cast<u32x4*>(bb + L::OFF_CT) =
            u32x4{0u, (unsigned)em.sel, (unsigned)(em.sel >> 32), nxt ? (unsigned)nxt->nfill : 0u};
    };

    Em cur, nxt;
    advance(cur);       // N >= 1: there is a first emission
    bool more = advance(nxt);
    {                   // prologue: the fill list of window 0 travels in buffer 1; emission 0 is staged
      int* fd = reinterpret_cast<int*>(smem + L::BUF + L::OFF_FD);
      if (lane < cur.nfill) {
        fd[2 * lane] = cur.fd_xc;
        fd[2 * lane + 1] = cur.fd_r0;
      }
      if (lane == 0) *reinterpret_cast<u32x4*>(smem + L::BUF + L::OFF_CT) = u32x4{0u, 0u, 0u, (unsigned)cur.nfill};
      barrier_lds();
      fill_share(fd, smem + L::OFF_WIN, cur.nfill);
      stage(cur, 0, more ? &nxt : nullptr);
      wait_vm0();
    }
#ifdef BEVR_GPROF
    unsigned long long gacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // iteration e: the row-block waves work on emission e; the producer stages emission e + 1 (`nxt`), fills its share of
    // window e + 1 and describes emission e + 2
    for (int e = 0;; ++e) {
      GPROF(p0);
      barrier_lds();      // emission e and window e are complete; buffer / window (e + 1) & 1 are free
      GPROF(p1);
      if (!more) {
        if (lane == 0) *reinterpret_cast<u32x4*>(smem + ((e + 1) & 1) * L::BUF + L::OFF_CT) = u32x4{4u, 0u, 0u, 0u};
        barrier_lds();
        break;
      }
      cur = nxt;          // emission e + 1
      {                   // the producer's columns of window e + 1; the fill list comes out of the lanes' registers
        char* wn = smem + L::OFF_WIN + ((e + 1) & 1) * L::WIN;
        for (int c = wave; c < cur.nfill; c += n_wave)
          fill_one(wn, c, __builtin_amdgcn_readlane(cur.fd_xc, c), __builtin_amdgcn_readlane(cur.fd_r0, c));
      }
      GPROF(p2);
      more = advance(nxt);
      stage(cur, e + 1, more ? &nxt : nullptr);
      GPROF(p3);
      wait_vm0();         // the K / V rows and the window columns have landed
      GPROF(p4);
      GPROF_ADD(4, p1 - p0);
      GPROF_ADD(6, p2 - p1);
      GPROF_ADD(1, p3 - p2);
      GPROF_ADD(3, p4 - p3);
      GPROF_ADD(5, 1);
    }
#ifdef BEVR_GPROF
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&bevr_prof_gather[i], gacc[i]);
#endif
    return;
  }

  // ---- row-block waves --------------------------------------------------------------------------------------
  const int blk0 = wave * NB;
  const int qb = prob / d.q_div;
  bf16x8 qf[NB];          // B operand of QK^T: Q[q][8 kg ..]
  f32x4 o_lo[NB], o_hi[NB];
  f32x4 negm[NB];         // minus the softmax reference: the accumulator start of every logit product
  f32x4 lacc[NB];         // row sum: a product of the ROUNDED weights with a ones operand (every register the same sum):
                          // O / l is then exact where one key dominates -- the backward's delta = dO . O relies on it
  float pmx[NB];          // the largest weight
  // byte offset of the lane's BEV row (first block) in ITS window column: k-groups 0, 2 read a key's column x, k-groups
  // 1, 3 column x + 1.  The second block's rows are QB dwords further: one ds_read2_b32 serves both blocks.
  const unsigned qoff0 = (unsigned)(blk0 * QB + li) * 4u + (unsigned)(kg & 1) * (PITCH * 4u);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int row = min(blk0 + nb, nblk - 1) * QB + li;
    const size_t mcol = (size_t)j * d.Sp + row;
    qf[nb] = __builtin_bit_cast(
        bf16x8, *reinterpret_cast<const u32x4*>(Q + ((((size_t)qb * d.heads + hd) * Mp + mcol) * 32 + 8 * kg) * 2));
    o_lo[nb] = o_hi[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float mr = EXACT ? 0.f : mref[(size_t)ph * Mp + mcol];
    negm[nb] = f32x4{-mr, -mr, -mr, -mr};
    lacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    pmx[nb] = 0.f;
  }
  const uint32_t one2 = Half<PREC>::pack2(1.0f, 1.0f);
  const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{one2, one2, one2, one2});
  const int t_off = (4 * kg + (li >> 2)) * 32 + (lane & 3) * 8;      // transposed reads of a [key][16] image
  const int k_off = li * 64 + 16 * (kg ^ (li >> 2));                 // this lane's K fragment in a 16-key block
  typedef const uint32_t __attribute__((address_space(3)))* lds_u32p;
  typedef const volatile uint32_t __attribute__((address_space(3)))* vlds_u32p;

  {   // prologue: window 0
    barrier_lds();
    const int nf0 = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<const u32x4*>(smem + L::BUF + L::OFF_CT)[0][3]);
    fill_share(reinterpret_cast<const int*>(smem + L::BUF + L::OFF_FD), smem + L::OFF_WIN, nf0);
    wait_vm0();
  }
  bool first = true;      // EXACT only
#ifdef BEVR_GPROF
  unsigned long long gacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long cprev = gprof_now();
#endif
  // the sweep over the emissions, instantiated for the distance (dwords) from the first block's window rows to the
  // second's: QB, or 0 for the wave whose second block lies past the column (it leaves that block out)
  auto sweep = [&](auto off1_c) {
  constexpr int OFF1 = decltype(off1_c)::value;
  constexpr int NBX = (NB == 2 && OFF1 == 0) ? 1 : NB;      // that wave works on its one block only
  for (int e = 0;; ++e) {
    GPROF(c0);
    barrier_lds();
    GPROF(c1);
    GPROF_ADD(0, c1 - c0);
    GPROF_ADD(4, c0 - cprev);
    GPROF_ADD(5, 1);
#ifdef BEVR_GPROF
    cprev = c1;
#endif
    const char* bb = smem + (e & 1) * L::BUF;
    const u32x4 ct = *reinterpret_cast<const u32x4*>(bb + L::OFF_CT);
    if (__builtin_amdgcn_readfirstlane((int)ct[0]) & 4) break;
    // this wave's share of the next window: in flight during the emission's work
    fill_share(reinterpret_cast<const int*>(bb + L::OFF_FD), smem + L::OFF_WIN + ((e + 1) & 1) * L::WIN,
               __builtin_amdgcn_readfirstlane((int)ct[3]));
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      const unsigned livem = (unsigned)__builtin_amdgcn_readfirstlane((int)(t ? ct[2] : ct[1]));
      // (no shortcut for a tile or sub-tile without live keys -- strip emissions only: its keys are masked like any
      // other; every accumulator update stays unconditional, which keeps the accumulators in place)
      const bf16x8 vlo = lds_tr8(bb + L::OFF_VLO + t * 1024 + t_off, 512);
      const bf16x8 vhi = lds_tr8(bb + L::OFF_VHI + t * 1024 + t_off, 512);
      u32x4 pw[NB];          // the tile's weights, packed: the B operand of PV
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int kb = 32 * t + 16 * s2;                 // first key of the sub-tile
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bb + L::OFF_K + kb * 64 + k_off));
        // W as the lanes hold it (w_image_pos): two 16-byte reads, no per-lane selection
        const bf16x8 wa0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bb + L::OFF_W + kb * 128 + lane * 16));
        const bf16x8 wa1 =
            __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bb + L::OFF_W + kb * 128 + 1024 + lane * 16));
        const int kq = 4 * (kg >> 1);
        // window addresses of keys kq .. kq + 3 (first bias product) and 8 + kq .. 8 + kq + 3 (second)
        const u32x4 oc0 = *reinterpret_cast<const u32x4*>(bb + L::OFF_OFF + (kb + kq) * 4);
        const u32x4 oc1 = *reinterpret_cast<const u32x4*>(bb + L::OFF_OFF + (kb + 8 + kq) * 4);
        // both row blocks' logit chains first (independent: their matrix products interleave), then the weights.  A wave
        // whose second block lies past the column computes it on the clamped rows and drops it in the epilogue.
        f32x4 sv[NB];
        u32x4 g0[NB], g1[NB];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          const lds_u32p p0 = (lds_u32p)(uintptr_t)(oc0[dd] + qoff0), p1 = (lds_u32p)(uintptr_t)(oc1[dd] + qoff0);
#pragma unroll
          for (int nb = 0; nb < NBX; ++nb) {
            // (volatile: kept from merging with the first block's read into a ds_read2_b32, whose register PAIR would have to
            // be moved apart into the two blocks' operands -- vector instructions are what this loop is short of)
            if (nb == 0) {
              g0[nb][dd] = p0[0];
              g1[nb][dd] = p1[0];
            } else {
              g0[nb][dd] = ((vlds_u32p)p0)[nb * OFF1];
              g1[nb][dd] = ((vlds_u32p)p1)[nb * OFF1];
            }
          }
        }
#pragma unroll
        for (int nb = 0; nb < NBX; ++nb) {
          sv[nb] = mfma16<PREC>(kf, qf[nb], negm[nb]);
          sv[nb] = mfma16<PREC>(wa0, __builtin_bit_cast(bf16x8, g0[nb]), sv[nb]);
          sv[nb] = mfma16<PREC>(wa1, __builtin_bit_cast(bf16x8, g1[nb]), sv[nb]);
        }
        if ((livem >> (16 * s2) & 0xffffu) != 0xffffu) {      // uniform: padding keys, or a strip emission
#pragma unroll
          for (int nb = 0; nb < NBX; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!(livem >> (16 * s2 + 4 * kg + r) & 1u)) sv[nb][r] = -1.0e30f;
        }
#pragma unroll
        for (int nb = 0; nb < NBX; ++nb) {
          if constexpr (EXACT) {
            // online reference: the maximum of the row's first keys, raised (with a rescale) when a later logit exceeds it
            const float tm = fmaxf(fmaxf(sv[nb][0], sv[nb][1]), fmaxf(sv[nb][2], sv[nb][3]));
            if (first || __any(tm > RAISE)) {
              float tx = fmaxf(tm, __shfl_xor(tm, 16));
              tx = fmaxf(tx, __shfl_xor(tx, 32));
              const float delta = first ? tx : fmaxf(tx, 0.f);
              const float al = first ? 1.0f : fast_exp2(-delta);
              negm[nb] -= delta;
              o_lo[nb] *= al;
              o_hi[nb] *= al;
              lacc[nb] *= al;
              pmx[nb] *= al;
              sv[nb] -= delta;
              if (s2 == 1) {       // the first sub-tile's weights of this tile are in pw already
                pw[nb][0] = Half<PREC>::pack2(Half<PREC>::lo(pw[nb][0]) * al, Half<PREC>::hi(pw[nb][0]) * al);
                pw[nb][1] = Half<PREC>::pack2(Half<PREC>::lo(pw[nb][1]) * al, Half<PREC>::hi(pw[nb][1]) * al);
              }
            }
          }
          const f32x2 pa = {fast_exp2(sv[nb][0]), fast_exp2(sv[nb][1])}, pb2 = {fast_exp2(sv[nb][2]), fast_exp2(sv[nb][3])};
          pw[nb][2 * s2] = Half<PREC>::pack2(pa[0], pa[1]);
          pw[nb][2 * s2 + 1] = Half<PREC>::pack2(pb2[0], pb2[1]);
          pmx[nb] = fmaxf(fmaxf(fmaxf(pmx[nb], pa[0]), pa[1]), fmaxf(pb2[0], pb2[1]));
        }
        if constexpr (EXACT) first = false;
      }
#pragma unroll
      for (int nb = 0; nb < NBX; ++nb) {
        const bf16x8 pb = __builtin_bit_cast(bf16x8, pw[nb]);
        o_lo[nb] = mfma16<PREC>(vlo, pb, o_lo[nb]);
        o_hi[nb] = mfma16<PREC>(vhi, pb, o_hi[nb]);
        lacc[nb] = mfma16<PREC>(ones, pb, lacc[nb]);
      }
    }
    GPROF(c2);
    GPROF_ADD(1, c2 - c1);
    wait_vm0();
    GPROF(c3);
    GPROF_ADD(2, c3 - c2);
  }
  };
  if constexpr (NB == 2) {
    if (blk0 + 1 < nblk) sweep(IntC<QB>{});
    else sweep(IntC<0>{});
  } else {
    sweep(IntC<0>{});
  }
#ifdef BEVR_GPROF
  if (lane == 0 && (wave == 0 || wave == 3)) for (int i = 0; i < 8; ++i) atomicAdd(&bevr_prof_gather[8 + (wave ? 8 : 0) + i], gacc[i]);
#endif

  // ---- epilogue: O[q][16 half + 4 kg ..], the two LSE planes; flag the column if a row's mass is not a healthy number ----
  bool bad = false;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (blk0 + nb >= nblk) continue;
    const float l = lacc[nb][0];
    float pm = fmaxf(pmx[nb], __shfl_xor(pmx[nb], 16));
    pm = fmaxf(pm, __shfl_xor(pm, 32));
    const int row = (blk0 + nb) * QB + li;
    if (!EXACT && row < d.S && !(l >= 7.9e-31f && l < 3.0e38f)) bad = true;
    // rows past the grid (zero Q rows of the last block) get their O and LSE like any other: the key-side backward
    // walks all Sp rows of a column and needs a finite LSE there; the rows no block covers take the last block's
    const float inv = 1.0f / l;
    const size_t mq = (size_t)ph * Mp + (size_t)j * d.Sp + row;
    float* orow = O + mq * 32;
    *reinterpret_cast<f32x4*>(orow + 4 * kg) = o_lo[nb] * inv;
    *reinterpret_cast<f32x4*>(orow + 16 + 4 * kg) = o_hi[nb] * inv;
    if (kg == 0) {
      const float lg = __log2f(l);
      const int n_copy = (blk0 + nb == nblk - 1) ? (d.Sp - row + QB - 1) / QB : 1;
      for (int k = 0; k < n_copy; ++k) {
        LSE[mq + k * QB] = lg - negm[nb][0];
        // plane 1: log2 of the row's largest softmax weight
        LSE[(size_t)n_ph * Mp + mq + k * QB] = __log2f(pm) - lg;
      }
    }
  }
  if constexpr (!EXACT) {
    if (__any(bad) && lane == 0) flags[ph * d.S + j] = 1;
  }
}

template <int PREC>
int launch(const bevr_attn_desc& d, const void* Q, const void* K, const void* V, const void* key_ws,
           const void* table_pk, const float* mref, float* O, float* LSE, int* flags, hipStream_t st) {
  typedef LdsG L;
  const int n_ph = d.n_prob * d.heads;
  const int grid = ((n_ph + 7) / 8) * 8 * d.S;
  const int nblk = (d.S + QB - 1) / QB;
  // a key's taps for all BEV rows of a column must fit one window column
  if (nblk * QB + 1 > ROWS || nblk > 14) return BEVR_E_SHAPE;
  const int nb = nblk <= 7 ? 1 : 2;      // row blocks per wave: at most 7 row-block waves + the producer
  const int n_cw = (nblk + nb - 1) / nb;
  const dim3 block(64 * (n_cw + 1));
#define BEVR_GATHER_LAUNCH(NB_, EX_, DMA_)                                                                           \
  hipLaunchKernelGGL((attn_gather_fwd_kernel<PREC, NB_, EX_, DMA_>), dim3(grid), block, L::TOTAL, st, d,            \
                     (const char*)Q, (const char*)K, (const char*)V, (const char*)key_ws, (const uint32_t*)table_pk, \
                     mref, O, LSE, flags)
  const bool dma = d.Hp >= ROWS;        // else (small problems): the window columns are copied through registers
  for (int ex = 0; ex < 2; ++ex) {       // static reference, then the exact pass over the flagged columns
    if (nb == 1) {
      if (dma) { if (ex) BEVR_GATHER_LAUNCH(1, true, true); else BEVR_GATHER_LAUNCH(1, false, true); }
      else { if (ex) BEVR_GATHER_LAUNCH(1, true, false); else BEVR_GATHER_LAUNCH(1, false, false); }
    } else {
      if (dma) { if (ex) BEVR_GATHER_LAUNCH(2, true, true); else BEVR_GATHER_LAUNCH(2, false, true); }
      else { if (ex) BEVR_GATHER_LAUNCH(2, true, false); else BEVR_GATHER_LAUNCH(2, false, false); }
    }
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
  }
#undef BEVR_GATHER_LAUNCH
  return BEVR_OK;
}

}  // namespace

extern "C" int bevr_attn_gather_fwd(const bevr_attn_desc* d, const void* Q, const void* K, const void* V,
                                    const void* key_ws, const void* table_pk, const float* mref, float* O, float* LSE,
                                    int* flags, void* stream) {
  int rc = bevr_check_desc(d);
  if (rc) return rc;
  if (!Q || !K || !V || !key_ws || !table_pk || !mref || !O || !LSE || !flags) return BEVR_E_NULL;
  if (!bevr_aligned16(Q) || !bevr_aligned16(K) || !bevr_aligned16(V) || !bevr_aligned16(O) || !bevr_aligned16(key_ws))
    return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (d->precision == BEVR_PREC_BF16) return launch<BEVR_PREC_BF16>(*d, Q, K, V, key_ws, table_pk, mref, O, LSE, flags, st);
  return BEVR_E_PRECISION;
}
