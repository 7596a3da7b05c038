// Bilinear key/value feature sampling (grid_sample, align_corners=True, zero padding) on a
// channels-last feature map, forward and backward.
// Replaces F.grid_sample at model/SCA_deform_attn.py:290-301 and model/TSA_deform_attn.py:210-217.
//
// HBM-bound gather: one thread owns 4 consecutive channels of one key, so the C/4 threads of a key
// read each of the 4 taps as one contiguous C*4-byte run (256 B at C = 64) and write one contiguous
// output row.  Backward scatters the same runs with float atomics (contiguous 256-B shapes, the form
// that runs at the full atomic rate) and reduces the position gradient over channels with wave
// shuffles.
#include "bevr_common.h"

namespace {

// feature element: float, or bf16 (the backbone's output dtype in the bf16 configurations: 8-byte tap reads instead of 16)
struct bf16_bits { unsigned short u; };
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_bits* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
               __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
}

struct Taps {
  int x0, y0;
  float fx, fy;
  bool vx0, vx1, vy0, vy1;
};

__device__ __forceinline__ Taps make_taps(float py, float px, int Hi, int Wi) {
  Taps t;
  float ix = (px + 1.0f) * 0.5f * (float)(Wi - 1);
  float iy = (py + 1.0f) * 0.5f * (float)(Hi - 1);
  float x0f = floorf(ix), y0f = floorf(iy);
  t.fx = ix - x0f;
  t.fy = iy - y0f;
  // clamp before the int conversion so NaN / huge positions cannot index out of range
  x0f = fminf(fmaxf(x0f, -2.0f), (float)Wi);
  y0f = fminf(fmaxf(y0f, -2.0f), (float)Hi);
  t.x0 = (int)x0f;
  t.y0 = (int)y0f;
  t.vx0 = t.x0 >= 0 && t.x0 < Wi;
  t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < Wi;
  t.vy0 = t.y0 >= 0 && t.y0 < Hi;
  t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < Hi;
  return t;
}

template <typename T>
__global__ __launch_bounds__(256) void sample_fwd_kernel(const T* __restrict__ feat,
                                                         const float* __restrict__ pos, float* __restrict__ out,
                                                         int nb, int Hi, int Wi, int C, int N) {
  const int c4n = C >> 2;
  const long long total = (long long)nb * N * c4n;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const long long kn = idx / c4n;  // b * N + n
    const int b = (int)(kn / N);
    const f32x2 p = *reinterpret_cast<const f32x2*>(pos + kn * 2);
    const Taps t = make_taps(p[0], p[1], Hi, Wi);
    const T* fb = feat + (size_t)b * Hi * Wi * C + c4 * 4;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v00 = (t.vy0 && t.vx0) ? load4(fb + ((size_t)t.y0 * Wi + t.x0) * C) : z;
    f32x4 v01 = (t.vy0 && t.vx1) ? load4(fb + ((size_t)t.y0 * Wi + t.x0 + 1) * C) : z;
    f32x4 v10 = (t.vy1 && t.vx0) ? load4(fb + ((size_t)(t.y0 + 1) * Wi + t.x0) * C) : z;
    f32x4 v11 = (t.vy1 && t.vx1) ? load4(fb + ((size_t)(t.y0 + 1) * Wi + t.x0 + 1) * C) : z;
    const float w00 = (1.f - t.fx) * (1.f - t.fy), w01 = t.fx * (1.f - t.fy);
    const float w10 = (1.f - t.fx) * t.fy, w11 = t.fx * t.fy;
    f32x4 r = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
    *reinterpret_cast<f32x4*>(out + kn * C + c4 * 4) = r;
  }
}

// Backward.  One key per group of G = C/4 lanes (G a power of two <= 64 is reduced with shuffles; otherwise the
// position gradient is added with atomics per thread).
//
// The feature gradient is a scatter of 4 contiguous C*4-byte runs per key (float atomics; contiguous 256-B shapes run
// at the full atomic rate).  One pattern does not: the projector pins every pillar point that falls outside a
// camera's image to pixel (0, 0) (model/bev_cmr_proj.py:76), so at the benchmark rig 66 % of the 100 000 keys of a
// view land, give or take the learned offset, on the same ~6 pixels of the feature map's top-left corner -- every
// workgroup's atomics on one 256-B row, the shape MI355X_MICROARCH.md measures 14x slower (17 ms per launch at B = 4).
// So the top-left HOT_R x HOT_C pixels are not scattered per key: each thread sums what its keys contribute to them
// in registers (the 2x2 tap weights as an outer product wy[row] * wx[col], no data-dependent register index), the
// workgroup reduces those sums over its 16 key slots through LDS, and ONE atomic per (hot pixel, channel) leaves the
// workgroup.  A workgroup stays inside one image so that the sums stay separable.
constexpr int HOT_R = 4, HOT_C = 2, SB_THREADS = 256;

template <bool HOT, typename T>
__global__ __launch_bounds__(SB_THREADS) void sample_bwd_kernel(const T* __restrict__ feat,
                                                                const float* __restrict__ pos,
                                                                const float* __restrict__ dout,
                                                                float* __restrict__ dfeat, float* __restrict__ dpos,
                                                                int nb, int Hi, int Wi, int C, int N, int pow2_group) {
  __shared__ float red[HOT ? SB_THREADS * HOT_R * HOT_C * 4 : 1];
  const int c4n = C >> 2;
  const int b = blockIdx.y;                       // one image per workgroup row
  const long long per_img = (long long)N * c4n;
  const long long stride = (long long)gridDim.x * blockDim.x;
  // padded up so that whole waves stay converged for the shuffles
  const long long padded = (per_img + 63) / 64 * 64;
  f32x4 hot[HOT ? HOT_R * HOT_C : 1];
#pragma unroll
  for (int p = 0; p < (HOT ? HOT_R * HOT_C : 1); ++p) hot[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const size_t img = (size_t)b * Hi * Wi * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < padded; idx += stride) {
    const bool live = idx < per_img;
    const long long id = live ? idx : per_img - 1;
    const int c4 = (int)(id % c4n);
    const long long kn = (long long)b * N + id / c4n;
    const f32x2 p = *reinterpret_cast<const f32x2*>(pos + kn * 2);
    const Taps t = make_taps(p[0], p[1], Hi, Wi);
    const size_t fo = img + c4 * 4;
    const T* fb = feat + fo;
    float* gb = dfeat + fo;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 g = live ? *reinterpret_cast<const f32x4*>(dout + kn * C + c4 * 4) : z;
    const bool b00 = t.vy0 && t.vx0, b01 = t.vy0 && t.vx1, b10 = t.vy1 && t.vx0, b11 = t.vy1 && t.vx1;
    const size_t o00 = ((size_t)t.y0 * Wi + t.x0) * C, o01 = o00 + C, o10 = o00 + (size_t)Wi * C, o11 = o10 + C;
    f32x4 v00 = b00 ? load4(fb + o00) : z;
    f32x4 v01 = b01 ? load4(fb + o01) : z;
    f32x4 v10 = b10 ? load4(fb + o10) : z;
    f32x4 v11 = b11 ? load4(fb + o11) : z;
    const float w00 = (1.f - t.fx) * (1.f - t.fy), w01 = t.fx * (1.f - t.fy);
    const float w10 = (1.f - t.fx) * t.fy, w11 = t.fx * t.fy;
    // which taps fall on the hot corner (those are summed in registers, the others scattered)
    bool h00 = false, h01 = false, h10 = false, h11 = false;
    if (HOT) {
      const bool ry0 = t.y0 >= 0 && t.y0 < HOT_R, ry1 = t.y0 + 1 >= 0 && t.y0 + 1 < HOT_R;
      const bool cx0 = t.x0 >= 0 && t.x0 < HOT_C, cx1 = t.x0 + 1 >= 0 && t.x0 + 1 < HOT_C;
      h00 = ry0 && cx0; h01 = ry0 && cx1; h10 = ry1 && cx0; h11 = ry1 && cx1;
      if (live) {
        // weight of hot pixel (r, c) for this key = wy[r] * wx[c]: the tap rows / columns it coincides with
#pragma unroll
        for (int r = 0; r < HOT_R; ++r) {
          const float wy = (t.y0 == r ? 1.f - t.fy : 0.f) + (t.y0 + 1 == r ? t.fy : 0.f);
#pragma unroll
          for (int c = 0; c < HOT_C; ++c) {
            const float wx = (t.x0 == c ? 1.f - t.fx : 0.f) + (t.x0 + 1 == c ? t.fx : 0.f);
            hot[r * HOT_C + c] += g * (wy * wx);
          }
        }
      }
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (b00 && !h00) atomicAdd(gb + o00 + k, g[k] * w00);
        if (b01 && !h01) atomicAdd(gb + o01 + k, g[k] * w01);
        if (b10 && !h10) atomicAdd(gb + o10 + k, g[k] * w10);
        if (b11 && !h11) atomicAdd(gb + o11 + k, g[k] * w11);
      }
    }
    // d out / d ix = (v01 - v00)(1 - fy) + (v11 - v10) fy ; d out / d iy = (v10 - v00)(1 - fx) + (v11 - v01) fx
    float gx = 0.f, gy = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gx += g[k] * ((v01[k] - v00[k]) * (1.f - t.fy) + (v11[k] - v10[k]) * t.fy);
      gy += g[k] * ((v10[k] - v00[k]) * (1.f - t.fx) + (v11[k] - v01[k]) * t.fx);
    }
    gx *= 0.5f * (float)(Wi - 1);
    gy *= 0.5f * (float)(Hi - 1);
    if (pow2_group) {
      for (int sh = c4n >> 1; sh > 0; sh >>= 1) {
        gx += __shfl_xor(gx, sh);
        gy += __shfl_xor(gy, sh);
      }
      if (live && c4 == 0) {
        f32x2 o = {gy, gx};
        *reinterpret_cast<f32x2*>(dpos + kn * 2) = o;
      }
    } else if (live) {
      atomicAdd(dpos + kn * 2, gy);
      atomicAdd(dpos + kn * 2 + 1, gx);
    }
  }
  if (HOT) {
    // HOT requires c4n | SB_THREADS (checked by the launcher): a thread keeps its channel quad c4 = tid % c4n over
    // the whole grid-stride loop, and the SB_THREADS / c4n threads that share a c4 are reduced here.
    const int tid = threadIdx.x;
    constexpr int NP = HOT_R * HOT_C;
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(red + ((size_t)p * SB_THREADS + tid) * 4) = hot[p];
    __syncthreads();
    const int slots = SB_THREADS / c4n;
    // (pixel p, channel quad c4): c4n * NP sums of `slots` partials each, dealt over the threads
    for (int u = tid; u < NP * c4n; u += SB_THREADS) {
      const int p = u / c4n, c4 = u % c4n;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int sl = 0; sl < slots; ++sl) a += *reinterpret_cast<const f32x4*>(red + ((size_t)p * SB_THREADS + sl * c4n + c4) * 4);
      const int r = p / HOT_C, c = p % HOT_C;
      if (r < Hi && c < Wi && (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f)) {
        float* gp = dfeat + img + ((size_t)r * Wi + c) * C + c4 * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(gp + k, a[k]);
      }
    }
  }
}

// Backward with an LDS patch (the benchmark path: C / 4 divides the workgroup size).
//
// The keys arrive in k-d order of their reference pixels (ops.split_key_order; the pinned keys cell-sorted behind
// them), so a run of consecutive keys touches a compact patch of the feature map -- at the benchmark rig 512 keys x 4
// taps fall on ~150 pixels.  A workgroup takes chunks of PCH_KEYS consecutive keys of one image: it finds the bounding
// box of the chunk's taps, accumulates every tap inside a window of up to PCH_BYTES of it in LDS, and flushes ONE
// global atomic per touched (pixel, channel): ~12x fewer global atomics than a scatter per key.  Taps outside the window
// (a chunk that straddles the image) are scattered directly; taps on the hot corner are summed in registers over all
// chunks of the workgroup as in sample_bwd_kernel<true>.
//
// The LDS cells are 64-bit FIXED POINT in units of gmax * 2^-40, gmax = the chunk's largest |dout| (found in the same
// pass as the bounding box): ds_add_f32 retires ~3 clk per active lane (193 clk per wave instruction,
// profiles/r02_lds_bench.txt) and owned 0.9 of the 3.2 ms of an SCA launch (racy read-modify-write probe: 2.35 ms);
// ds_add_u64 takes 6.3 clk.  A tap value is at most gmax (bilinear weights <= 1) and a cell collects at most
// 4 * PCH_KEYS of them: |cell| < 2^10 * 2^40; a contribution is rounded to 2^-40 of gmax -- far below float rounding --
// and the integer sum is order-independent.
constexpr int PCH_KEYS = 256, PCH_BYTES = 48 * 1024, PCH_WMAX = 32;

// x / gmax * 2^40 as (hi: signed 32 bits, lo: unsigned 32 bits), t = x * 2^8 / gmax in [-256, 256]
__device__ __forceinline__ unsigned long long to_fixed40(float t) {
  const float h = floorf(t);
  // t - h lies in [0, 1) but ROUNDS to 1.0f for a tiny negative t: clamp below 1 (a float -> unsigned conversion of
  // 2^32 is out of range: undefined, harmless only because v_cvt_u32_f32 saturates)
  const unsigned lo = (unsigned)(fminf(t - h, 0x1.fffffep-1f) * 4294967296.0f);
  return ((unsigned long long)(unsigned)(int)h << 32) | lo;
}

template <typename T>
__global__ __launch_bounds__(SB_THREADS) void sample_bwd_patch_kernel(const T* __restrict__ feat,
                                                                      const float* __restrict__ pos,
                                                                      const float* __restrict__ dout,
                                                                      float* __restrict__ dfeat, float* __restrict__ dpos,
                                                                      int nb, int Hi, int Wi, int C, int N, int pow2_group) {
  __shared__ __attribute__((aligned(16))) unsigned long long patch64[PCH_BYTES / 8];   // the window (fixed point)
  float* patch = reinterpret_cast<float*>(patch64);                                     // at the end: the hot corner's partials
  __shared__ int s_box[SB_THREADS / 64][5];
  const int c4n = C >> 2, tid = threadIdx.x;
  const int b = blockIdx.y;
  const int slots = SB_THREADS / c4n, slot = tid / c4n, c4 = tid % c4n;
  const int pix_cap = PCH_BYTES / (C * 8);
  const size_t img = (size_t)b * Hi * Wi * C;
  const T* fb = feat + img + c4 * 4;
  float* gimg = dfeat + img;
  f32x4 hot[HOT_R * HOT_C];
#pragma unroll
  for (int p = 0; p < HOT_R * HOT_C; ++p) hot[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_chunk = (N + PCH_KEYS - 1) / PCH_KEYS;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  for (int chunk = blockIdx.x; chunk < n_chunk; chunk += gridDim.x) {
    const int k0 = chunk * PCH_KEYS, nk = min(PCH_KEYS, N - k0);
    // ---- bounding box of the chunk's taps (inside the image, outside the hot corner) ----
    int xlo = 1 << 30, xhi = -1, ylo = 1 << 30, yhi = -1;
    for (int kk = tid; kk < nk; kk += SB_THREADS) {
      const f32x2 p = *reinterpret_cast<const f32x2*>(pos + ((size_t)b * N + k0 + kk) * 2);
      const Taps t = make_taps(p[0], p[1], Hi, Wi);
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int x = t.x0 + dx, y = t.y0 + dy;
          const bool in = x >= 0 && x < Wi && y >= 0 && y < Hi && !(y < HOT_R && x < HOT_C);
          if (in) { xlo = min(xlo, x); xhi = max(xhi, x); ylo = min(ylo, y); yhi = max(yhi, y); }
        }
    }
    float gmax = 0.f;   // the chunk's largest |dout| (the main loop reads the rows again: L2 hits)
    for (int u = tid; u < nk * c4n; u += SB_THREADS) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(dout + ((size_t)b * N + k0 + u / c4n) * C + (u % c4n) * 4);
      gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(g4[0]), fabsf(g4[1])), fmaxf(fabsf(g4[2]), fabsf(g4[3]))));
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
      xlo = min(xlo, __shfl_xor(xlo, sh)); xhi = max(xhi, __shfl_xor(xhi, sh));
      ylo = min(ylo, __shfl_xor(ylo, sh)); yhi = max(yhi, __shfl_xor(yhi, sh));
      gmax = fmaxf(gmax, __shfl_xor(gmax, sh));
    }
    if ((tid & 63) == 0) {
      s_box[tid >> 6][0] = xlo; s_box[tid >> 6][1] = xhi; s_box[tid >> 6][2] = ylo; s_box[tid >> 6][3] = yhi;
      s_box[tid >> 6][4] = __builtin_bit_cast(int, gmax);
    }
    __syncthreads();   // also: the previous chunk's flush is complete
#pragma unroll
    for (int w = 0; w < SB_THREADS / 64; ++w) {
      xlo = min(xlo, s_box[w][0]); xhi = max(xhi, s_box[w][1]); ylo = min(ylo, s_box[w][2]); yhi = max(yhi, s_box[w][3]);
      gmax = fmaxf(gmax, __builtin_bit_cast(float, s_box[w][4]));
    }
    // NaN / inf in dout: no finite unit -- every tap of the chunk takes the direct scatter (pw = 0)
    const bool unit_ok = gmax > 0.f && gmax < 3.0e38f;
    const float to_fix = unit_ok ? 256.0f / gmax : 0.f, from_fix = unit_ok ? gmax * 0x1p-40f : 0.f;
    const int pw = (xhi >= xlo && unit_ok) ? min(xhi - xlo + 1, PCH_WMAX) : 0;
    const int ph = pw > 0 ? min(yhi - ylo + 1, pix_cap / pw) : 0;
    const int n_el = ph * pw * C;
    for (int u = tid * 2; u < n_el; u += SB_THREADS * 2) *reinterpret_cast<u32x4*>(patch64 + u) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- the chunk's keys, `slots` at a time (whole waves stay converged for the shuffles) ----
    const int nk_pad = (nk + slots - 1) / slots * slots;
    for (int kk = slot; kk < nk_pad; kk += slots) {
      const bool live = kk < nk;
      const size_t kn = (size_t)b * N + k0 + (live ? kk : nk - 1);
      const f32x2 p = *reinterpret_cast<const f32x2*>(pos + kn * 2);
      const Taps t = make_taps(p[0], p[1], Hi, Wi);
      const f32x4 g = live ? *reinterpret_cast<const f32x4*>(dout + kn * C + c4 * 4) : z;
      const bool bv[4] = {t.vy0 && t.vx0, t.vy0 && t.vx1, t.vy1 && t.vx0, t.vy1 && t.vx1};
      const float wv[4] = {(1.f - t.fx) * (1.f - t.fy), t.fx * (1.f - t.fy), (1.f - t.fx) * t.fy, t.fx * t.fy};
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t o = ((size_t)(t.y0 + (q >> 1)) * Wi + t.x0 + (q & 1)) * C;
        v[q] = bv[q] ? load4(fb + o) : z;
      }
      if (live) {
        // hot corner: weight of hot pixel (r, c) = wy[r] * wx[c] (no data-dependent register index)
#pragma unroll
        for (int r = 0; r < HOT_R; ++r) {
          const float wy = (t.y0 == r ? 1.f - t.fy : 0.f) + (t.y0 + 1 == r ? t.fy : 0.f);
#pragma unroll
          for (int c = 0; c < HOT_C; ++c) {
            const float wx = (t.x0 == c ? 1.f - t.fx : 0.f) + (t.x0 + 1 == c ? t.fx : 0.f);
            hot[r * HOT_C + c] += g * (wy * wx);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int x = t.x0 + (q & 1), y = t.y0 + (q >> 1);
          if (!bv[q] || (y < HOT_R && x < HOT_C)) continue;
          const int px = x - xlo, py = y - ylo;
          if (px >= 0 && px < pw && py >= 0 && py < ph) {
            // element (pixel, c4, k) at pixel * C + ((k * c4n + c4 + 16 * pixel) mod C): the key slots of a wave land on
            // different bank groups when their pixels differ
            const int pix = py * pw + px;
            unsigned long long* pp = patch64 + pix * C;
            const int rot = c4 + 16 * pix;
            const float wt = wv[q] * to_fix;
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(pp + ((k * c4n + rot) % C), to_fixed40(g[k] * wt));
          } else {
            float* gp = gimg + ((size_t)y * Wi + x) * C + c4 * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(gp + k, g[k] * wv[q]);
          }
        }
      }
      float gx = 0.f, gy = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gx += g[k] * ((v[1][k] - v[0][k]) * (1.f - t.fy) + (v[3][k] - v[2][k]) * t.fy);
        gy += g[k] * ((v[2][k] - v[0][k]) * (1.f - t.fx) + (v[3][k] - v[1][k]) * t.fx);
      }
      gx *= 0.5f * (float)(Wi - 1);
      gy *= 0.5f * (float)(Hi - 1);
      if (pow2_group) {
        for (int sh = c4n >> 1; sh > 0; sh >>= 1) {
          gx += __shfl_xor(gx, sh);
          gy += __shfl_xor(gy, sh);
        }
        if (live && c4 == 0) *reinterpret_cast<f32x2*>(dpos + kn * 2) = f32x2{gy, gx};
      } else if (live) {
        atomicAdd(dpos + kn * 2, gy);
        atomicAdd(dpos + kn * 2 + 1, gx);
      }
    }
    __syncthreads();
    // ---- flush the window: one atomic per touched (pixel, channel), 256-B rows ----
    for (int u = tid; u < ph * pw * c4n; u += SB_THREADS) {
      const int pix = u / c4n, cc = u % c4n;
      const unsigned long long* pp = patch64 + pix * C;
      const int rot = cc + 16 * pix;
      float a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = (float)(long long)pp[(k * c4n + rot) % C] * from_fix;   // the cell converted whole
      if (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f) {
        float* gp = gimg + ((size_t)(ylo + pix / pw) * Wi + xlo + pix % pw) * C + cc * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(gp + k, a[k]);
      }
    }
  }
  // ---- hot corner: reduce the workgroup's partials (the threads that share a channel quad) and flush ----
  __syncthreads();
  constexpr int NP = HOT_R * HOT_C;
  static_assert(NP * SB_THREADS * 16 <= PCH_BYTES, "hot partials reuse the patch");
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(patch + ((size_t)p * SB_THREADS + tid) * 4) = hot[p];
  __syncthreads();
  for (int u = tid; u < NP * c4n; u += SB_THREADS) {
    const int p = u / c4n, cc = u % c4n;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < slots; ++sl) a += *reinterpret_cast<const f32x4*>(patch + ((size_t)p * SB_THREADS + sl * c4n + cc) * 4);
    const int r = p / HOT_C, c = p % HOT_C;
    if (r < Hi && c < Wi && (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f)) {
      float* gp = gimg + ((size_t)r * Wi + c) * C + cc * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(gp + k, a[k]);
    }
  }
}

int grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;  // grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T>
int sample_fwd(const T* feat, const float* pos, float* out, int nb, int Hi, int Wi, int C, int N, void* stream) {
  if (!feat || !pos || !out) return BEVR_E_NULL;
  if (nb <= 0 || Hi < 2 || Wi < 2 || N <= 0 || C <= 0 || (C & 3) || C > 1024) return BEVR_E_SHAPE;
  if (!bevr_aligned16(feat) || !bevr_aligned16(out) || (reinterpret_cast<uintptr_t>(pos) & 7)) return BEVR_E_ALIGN;
  long long total = (long long)nb * N * (C >> 2);
  hipLaunchKernelGGL(sample_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, feat, pos, out,
                     nb, Hi, Wi, C, N);
  return (int)hipGetLastError();
}

template <typename T>
int sample_bwd(const T* feat, const float* pos, const float* dout, float* dfeat, float* dpos, int nb, int Hi, int Wi,
               int C, int N, void* stream) {
  if (!feat || !pos || !dout || !dfeat || !dpos) return BEVR_E_NULL;
  if (nb <= 0 || Hi < 2 || Wi < 2 || N <= 0 || C <= 0 || (C & 3) || C > 1024) return BEVR_E_SHAPE;
  if (!bevr_aligned16(feat) || !bevr_aligned16(dout) || !bevr_aligned16(dfeat) ||
      (reinterpret_cast<uintptr_t>(pos) & 7) || (reinterpret_cast<uintptr_t>(dpos) & 7))
    return BEVR_E_ALIGN;
  const int c4n = C >> 2;
  const int pow2 = (c4n <= 64 && (c4n & (c4n - 1)) == 0) ? 1 : 0;
  if (!pow2) {
    hipError_t e = hipMemsetAsync(dpos, 0, (size_t)nb * N * 2 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  // the LDS cells are 8-byte fixed point: the widest window row (PCH_WMAX pixels) must fit
  const bool patch_ok = (SB_THREADS % c4n) == 0 && C * 8 * PCH_WMAX <= PCH_BYTES;
  if (patch_ok) {
    // chunks of consecutive keys; a workgroup takes several (the hot corner's partials are reduced once per workgroup)
    const int n_chunk = (N + PCH_KEYS - 1) / PCH_KEYS;
    long long gx = n_chunk;
    const long long want = (256LL * 12 + nb - 1) / nb;
    if (gx > want) gx = want;
    hipLaunchKernelGGL(sample_bwd_patch_kernel<T>, dim3((unsigned)gx, nb), dim3(SB_THREADS), 0, (hipStream_t)stream,
                       feat, pos, dout, dfeat, dpos, nb, Hi, Wi, C, N, pow2);
  } else {
    // one image per workgroup row, grid-stride over its keys, every tap scattered
    const long long per_img = (long long)N * c4n;
    long long gx = (per_img + SB_THREADS - 1) / SB_THREADS;
    const long long want = (256LL * 16 + nb - 1) / nb;
    if (gx > want) gx = want;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((sample_bwd_kernel<false, T>), dim3((unsigned)gx, nb), dim3(SB_THREADS), 0, (hipStream_t)stream,
                       feat, pos, dout, dfeat, dpos, nb, Hi, Wi, C, N, pow2);
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int bevr_sample_fwd(const float* feat, const float* pos, float* out, int nb, int Hi, int Wi, int C,
                               int N, void* stream) {
  return sample_fwd(feat, pos, out, nb, Hi, Wi, C, N, stream);
}
extern "C" int bevr_sample_bwd(const float* feat, const float* pos, const float* dout, float* dfeat, float* dpos,
                               int nb, int Hi, int Wi, int C, int N, void* stream) {
  return sample_bwd(feat, pos, dout, dfeat, dpos, nb, Hi, Wi, C, N, stream);
}
// the same on a bf16 feature map (raw bits); outputs and gradients stay float
extern "C" int bevr_sample_fwd_bf16(const void* feat, const float* pos, float* out, int nb, int Hi, int Wi, int C,
                                    int N, void* stream) {
  return sample_fwd(static_cast<const bf16_bits*>(feat), pos, out, nb, Hi, Wi, C, N, stream);
}
extern "C" int bevr_sample_bwd_bf16(const void* feat, const float* pos, const float* dout, float* dfeat, float* dpos,
                                    int nb, int Hi, int Wi, int C, int N, void* stream) {
  return sample_bwd(static_cast<const bf16_bits*>(feat), pos, dout, dfeat, dpos, nb, Hi, Wi, C, N, stream);
}
