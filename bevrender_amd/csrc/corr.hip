// Ground <-> aerial correlation: D = 2 - 2 * cam map^T on raw or L2-normalised rows, its backward,
// and the rank-of-diagonal count of Trainer.get_recall.
// Replaces train.py:554 (np.matmul on the host) and the pairwise-distance matrix inside the
// retrieval losses (loss/contrastive_loss.py:10-19, loss/lift_loss.py:13-22).
//
// At training batch sizes (n, m <= 32) with E = C*S*S (2.56 M at S = 200) this is HBM-bound: every
// embedding element is read exactly once.  Each workgroup owns a slice of E, keeps an 8x8 block of
// partial dot products per thread in registers, reduces over the workgroup and adds 64 floats with
// atomics.  (A validation-size Gram with n in the thousands is a real GEMM and belongs on MFMA: a
// later-round row, see DESIGN.md.)
#include <stdlib.h>
#include "bevr_common.h"

namespace {

constexpr int TB = 8;        // rows per register block
constexpr int ECHUNK = 4096; // embedding elements per workgroup (256 threads x 4 x 4)

__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ sq, int rows,
                                                      int E) {
  // grid (chunks, rows): partial sum of squares, atomically added
  const int r = blockIdx.y;
  const long long e0 = (long long)blockIdx.x * ECHUNK;
  float acc = 0.f;
  for (int k = threadIdx.x * 4; k < ECHUNK; k += 1024) {
    long long e = e0 + k;
    if ((E & 3) == 0 && e + 3 < E) {   // rows are 16-byte aligned only when E % 4 == 0
      f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * E + e);
      acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (int q = 0; q < 4; ++q)
        if (e + q < E) { float v = x[(size_t)r * E + e + q]; acc += v * v; }
    }
  }
  for (int sh = 32; sh > 0; sh >>= 1) acc += __shfl_xor(acc, sh);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sq + r, part[0] + part[1] + part[2] + part[3]);
}

__global__ void finish_norm_kernel(float* sq, int rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // F.normalize: x / max(||x||, 1e-12)
  if (i < rows) sq[i] = 1.0f / fmaxf(sqrtf(sq[i]), 1e-12f);
}

// dots[i][j] += sum_{e in chunk} cam[i][e] * map[j][e]
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ cam, const float* __restrict__ map,
                                                   float* __restrict__ dots, int n, int m, int E) {
  const int ib = blockIdx.y * TB, jb = blockIdx.z * TB;
  const long long e0 = (long long)blockIdx.x * ECHUNK;
  float acc[TB][TB];
#pragma unroll
  for (int a = 0; a < TB; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = 0.f;
  const bool vec_ok = (E & 3) == 0;
  for (int k = threadIdx.x * 4; k < ECHUNK; k += 1024) {
    const long long e = e0 + k;
    if (e >= E) break;
    f32x4 a[TB], b[TB];
#pragma unroll
    for (int r = 0; r < TB; ++r) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      a[r] = z; b[r] = z;
      if (vec_ok && e + 3 < E) {
        if (ib + r < n) a[r] = *reinterpret_cast<const f32x4*>(cam + (size_t)(ib + r) * E + e);
        if (jb + r < m) b[r] = *reinterpret_cast<const f32x4*>(map + (size_t)(jb + r) * E + e);
      } else {
        for (int q = 0; q < 4; ++q) {
          if (e + q < E && ib + r < n) a[r][q] = cam[(size_t)(ib + r) * E + e + q];
          if (e + q < E && jb + r < m) b[r][q] = map[(size_t)(jb + r) * E + e + q];
        }
      }
    }
#pragma unroll
    for (int x = 0; x < TB; ++x)
#pragma unroll
      for (int y = 0; y < TB; ++y)
        acc[x][y] += a[x][0] * b[y][0] + a[x][1] * b[y][1] + a[x][2] * b[y][2] + a[x][3] * b[y][3];
  }
  __shared__ float red[4][TB * TB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int x = 0; x < TB; ++x)
#pragma unroll
    for (int y = 0; y < TB; ++y) {
      float v = acc[x][y];
      for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
      if (lane == 0) red[wave][x * TB + y] = v;
    }
  __syncthreads();
  if (threadIdx.x < TB * TB) {
    const int x = threadIdx.x / TB, y = threadIdx.x % TB;
    if (ib + x < n && jb + y < m)
      atomicAdd(dots + (size_t)(ib + x) * m + jb + y,
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

__global__ void finish_dist_kernel(float* D, const float* inc, const float* inm, int n, int m, int normalize) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * m) return;
  float s = normalize ? inc[idx / m] * inm[idx % m] : 1.0f;
  D[idx] = 2.0f - 2.0f * D[idx] * s;
}

// dX[i][e] = sum_j W[i][j] * Y[j][e] (+ optional  coef[i] * X[i][e]);  used for both sides of the backward.
//   raw:        dcam_i = -2 sum_j dD_ij map_j
//   normalised: dcam_i = -2/|c_i| ( sum_j dD_ij m^_j  -  (sum_j dD_ij <c^_i, m^_j>) c^_i )
__global__ __launch_bounds__(256) void corr_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                       const float* __restrict__ W, const float* __restrict__ Dm,
                                                       const float* __restrict__ inx, const float* __restrict__ iny,
                                                       float* __restrict__ dX, int nx, int ny, int E, int transposed,
                                                       int normalize) {
  // grid (E chunks of 1024, nx rows)
  const int i = blockIdx.y;
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= E) return;
  float accv[4] = {0.f, 0.f, 0.f, 0.f};
  float self = 0.f;
  for (int j = 0; j < ny; ++j) {
    const size_t wi = transposed ? (size_t)j * nx + i : (size_t)i * ny + j;
    float w = W[wi];
    if (normalize) {
      self += w * (2.0f - Dm[wi]) * 0.5f;  // <x^_i, y^_j>
      w *= iny[j];
    }
    for (int q = 0; q < 4; ++q)
      if (e + q < E) accv[q] += w * Y[(size_t)j * E + e + q];
  }
  const float sx = normalize ? inx[i] : 1.0f;
  for (int q = 0; q < 4; ++q)
    if (e + q < E) {
      float v = accv[q];
      if (normalize) v -= self * X[(size_t)i * E + e + q] * sx;
      dX[(size_t)i * E + e + q] = -2.0f * sx * v;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// One-pass forms for training-size matrices (n, m <= CORR_MAX_ROWS): a workgroup owns a slice of E and visits every
// (8 x 8) block of the matrix itself, so each embedding element leaves HBM ONCE (the re-reads of a slice by its own
// workgroup are L1 / L2 hits); 16-byte loads throughout.  Round 4's kernels launched one workgroup per (slice, row block,
// column block) -- every row read (n / 8 + m / 8) / 2 times -- plus two row-norm passes, and the backward re-read all of Y
// for every output row with scalar loads: 635 us per call for 246 MB of compulsory bytes.
constexpr int CORR_MAX_ROWS = 64;

// dots[i][j] += <x_i, y_j>: the Gram of a training batch IS a (n x E) x (E x m) GEMM with a tiny output -- on the matrix
// cores in f32 (v_mfma_f32_16x16x4_f32: the embeddings are float and the distances feed margins; no operand rounding), with
// the whole n x m result in accumulator registers across a wave's grid-stride walk over E, so that every embedding
// element is read from HBM once and only ONE reduction per workgroup leaves (256 atomics per 16 x 16 tile and workgroup).
// Lane (r = lane & 15, kg = lane >> 4) loads 16 bytes of row 16 f + r at element e0 + 4 kg: a wave instruction covers 64
// contiguous bytes of each of 16 rows; element q of the vector is contraction index 4 kg + q of MFMA number q -- the same
// lane layout serves the A operand (rows of X) and the B operand (rows of Y), so for X == Y the registers are shared.
// With sqx / sqy: the rows' squared norms in the same pass (F.normalize's denominators).
typedef __attribute__((ext_vector_type(4))) float acc4;

template <int FA, int FB, int NW>
__global__ __launch_bounds__(64 * NW) void gram_mfma_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                        float* __restrict__ dots, float* __restrict__ sqx,
                                                        float* __restrict__ sqy, int n, int m, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const bool same = X == Y && n == m;
  acc4 acc[FA][FB];
  float nx[FA], ny[FB];
#pragma unroll
  for (int fa = 0; fa < FA; ++fa) {
    nx[fa] = 0.f;
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) acc[fa][fb] = acc4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) ny[fb] = 0.f;
  const long long stride = (long long)gridDim.x * NW * 32;      // elements per grid sweep: every wave takes 32 per step
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (long long e0 = ((long long)blockIdx.x * NW + wave) * 32; e0 < E; e0 += stride) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                // two 16-element steps: both halves of a 128-byte line
      const long long e = e0 + 16 * u + 4 * kg;                  // E % 4 == 0: the lane's 4 elements are in or out together
      f32x4 a[FA], b[FB];
#pragma unroll
      for (int fa = 0; fa < FA; ++fa)
        a[fa] = (e < E && 16 * fa + r < n) ? *reinterpret_cast<const f32x4*>(X + (size_t)(16 * fa + r) * E + e) : z;
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        if (same) b[fb] = a[fb < FA ? fb : 0];
        else b[fb] = (e < E && 16 * fb + r < m) ? *reinterpret_cast<const f32x4*>(Y + (size_t)(16 * fb + r) * E + e) : z;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb)
            acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[fa][q], b[fb][q], acc[fa][fb], 0, 0, 0);
      if (sqx) {
#pragma unroll
        for (int fa = 0; fa < FA; ++fa) nx[fa] += a[fa][0] * a[fa][0] + a[fa][1] * a[fa][1] + a[fa][2] * a[fa][2] + a[fa][3] * a[fa][3];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) ny[fb] += b[fb][0] * b[fb][0] + b[fb][1] * b[fb][1] + b[fb][2] * b[fb][2] + b[fb][3] * b[fb][3];
      }
    }
  }
  // ---- one reduction per workgroup and 16 x 16 tile: the 4 waves through LDS, then atomics ----
  __shared__ float red[NW][256];
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
#pragma unroll
      for (int k = 0; k < 4; ++k)   // accumulator element k of lane: row i = 4 (lane >> 4) + k, column j = lane & 15
        red[wave][(4 * kg + k) * 16 + r] = acc[fa][fb][k];
      __syncthreads();
      if (threadIdx.x < 256) {
        const int k = threadIdx.x, i = 16 * fa + k / 16, j = 16 * fb + k % 16;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][k];
        if (i < n && j < m) atomicAdd(dots + (size_t)i * m + j, v);
      }
      __syncthreads();
    }
  if (sqx) {
    // squared row norms: the lane's partial over its k-groups, then the waves
    for (int f = 0; f < FA + FB; ++f) {
      float v = 0.f;
#pragma unroll
      for (int fa = 0; fa < FA; ++fa) if (f == fa) v = nx[fa];
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) if (f == FA + fb) v = ny[fb];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kg == 0) red[wave][r] = v;
      __syncthreads();
      if (threadIdx.x < 16) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][threadIdx.x];
        const int row = 16 * (f < FA ? f : f - FA) + threadIdx.x;
        if (f < FA) { if (row < n) atomicAdd(sqx + row, t); }
        else if (row < m) atomicAdd(sqy + row, t);
      }
      __syncthreads();
    }
  }
}

// D = 2 - 2 dots / (|x_i| |y_j|); the squared norms become F.normalize's 1 / max(|x|, 1e-12) on the way
__global__ void finish_dist2_kernel(float* D, float* sqx, float* sqy, int n, int m, int normalize) {
  __shared__ float ix[CORR_MAX_ROWS], iy[CORR_MAX_ROWS];
  if (normalize) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) ix[i] = 1.0f / fmaxf(sqrtf(sqx[i]), 1e-12f);
    for (int j = threadIdx.x; j < m; j += blockDim.x) iy[j] = 1.0f / fmaxf(sqrtf(sqy[j]), 1e-12f);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < n * m; idx += blockDim.x)
    D[idx] = 2.0f - 2.0f * D[idx] * (normalize ? ix[idx / m] * iy[idx % m] : 1.0f);
  __syncthreads();
  if (normalize) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) sqx[i] = ix[i];
    for (int j = threadIdx.x; j < m; j += blockDim.x) sqy[j] = iy[j];
  }
}

// Both sides of the backward over one slice of E (see corr_bwd_kernel for the formulas):
//   dX[i] = -2 sx_i ( sum_j W[i][j] sy_j y_j  -  self_i sx_i x_i ),   self_i = sum_j W[i][j] <x^_i, y^_j>
//   dY[j] = -2 sy_j ( sum_i W[i][j] sx_i x_i  -  selfT_j sy_j y_j )
// (sx = sy = 1, self = 0 on raw rows).  SYM: X and Y are the SAME buffer (the retrieval losses correlate cat(cam, map)
// with itself): the two sides are summed, dX[i] = dcam[i] + dmap[i], one read and one write of the embedding.
// The weights (with the other side's norm folded in) sit in LDS, transposed so that a register block's 8 weights of one
// contraction index are two 16-byte broadcasts.
template <bool SYM>
__global__ __launch_bounds__(256) void corr_bwd_slice_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                             const float* __restrict__ W, const float* __restrict__ Dm,
                                                             const float* __restrict__ inx, const float* __restrict__ iny,
                                                             float* __restrict__ dX, float* __restrict__ dY, int nx, int ny,
                                                             int E, int normalize) {
  // wx[j][i] = W[i][j] * sy_j (+ W[j][i] * sy_j if SYM: nx == ny), padded to 8 in i;  wy[i][j] = W[i][j] * sx_i
  __shared__ __attribute__((aligned(16))) float wx[CORR_MAX_ROWS * CORR_MAX_ROWS], wy[CORR_MAX_ROWS * CORR_MAX_ROWS];
  __shared__ float selfx[CORR_MAX_ROWS], selfy[CORR_MAX_ROWS];
  const int nxp = (nx + TB - 1) / TB * TB, nyp = (ny + TB - 1) / TB * TB;
  for (int k = threadIdx.x; k < nxp * nyp; k += 256) {
    const int j = k / nxp, i = k % nxp;     // wx index: [j][i]
    float w = 0.f;
    if (i < nx && j < ny) {
      w = W[(size_t)i * ny + j];
      if (SYM) w += W[(size_t)j * ny + i];
      if (normalize) w *= iny[j];
    }
    wx[k] = w;
    if (!SYM) {
      const int i2 = k / nyp, j2 = k % nyp;   // wy index: [i][j]
      float w2 = 0.f;
      if (i2 < nx && j2 < ny) w2 = W[(size_t)i2 * ny + j2] * (normalize ? inx[i2] : 1.0f);
      wy[k] = w2;
    }
  }
  for (int i = threadIdx.x; i < CORR_MAX_ROWS; i += 256) {
    float sx_ = 0.f, sy_ = 0.f;
    if (normalize) {
      if (i < nx)
        for (int j = 0; j < ny; ++j) {
          sx_ += W[(size_t)i * ny + j] * (2.0f - Dm[(size_t)i * ny + j]) * 0.5f;
          if (SYM) sx_ += W[(size_t)j * ny + i] * (2.0f - Dm[(size_t)j * ny + i]) * 0.5f;
        }
      if (!SYM && i < ny)
        for (int k = 0; k < nx; ++k) sy_ += W[(size_t)k * ny + i] * (2.0f - Dm[(size_t)k * ny + i]) * 0.5f;
    }
    selfx[i] = sx_;
    selfy[i] = sy_;
  }
  __syncthreads();
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= E) return;
  auto side = [&](const float* __restrict__ Xs, const float* __restrict__ Ys, const float* __restrict__ wl,
                  const float* __restrict__ selfs, const float* __restrict__ inxs, float* __restrict__ dXs, int nxs, int nys,
                  int nxsp) {
    for (int ib = 0; ib < nxs; ib += TB) {
      f32x4 acc[TB];
#pragma unroll
      for (int r = 0; r < TB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < nys; ++j) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(Ys + (size_t)j * E + e);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (size_t)j * nxsp + ib);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + (size_t)j * nxsp + ib + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] += yv * w0[r]; acc[4 + r] += yv * w1[r]; }
      }
#pragma unroll
      for (int r = 0; r < TB; ++r) {
        if (ib + r < nxs) {
          const float sx = normalize ? inxs[ib + r] : 1.0f;
          f32x4 v = acc[r];
          if (normalize) v -= *reinterpret_cast<const f32x4*>(Xs + (size_t)(ib + r) * E + e) * (selfs[ib + r] * sx);
          *reinterpret_cast<f32x4*>(dXs + (size_t)(ib + r) * E + e) = v * (-2.0f * sx);
        }
      }
    }
  };
  side(X, Y, wx, selfx, inx, dX, nx, ny, nxp);
  if (!SYM) side(Y, X, wy, selfy, iny, dY, ny, nx, nyp);
}

__global__ void recall_rank_kernel(const float* __restrict__ D, int32_t* __restrict__ rank, int n) {
  const int k = blockIdx.x;  // column
  const float gt = D[(size_t)k * n + k];
  int cnt = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) cnt += D[(size_t)i * n + k] < gt ? 1 : 0;
  for (int sh = 32; sh > 0; sh >>= 1) cnt += __shfl_xor(cnt, sh);
  __shared__ int part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) rank[k] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace

extern "C" int bevr_corr_fwd(const float* cam, const float* map, float* D, float* inv_norm_cam, float* inv_norm_map,
                             int n, int m, int E, int normalize, void* stream) {
  if (!cam || !map || !D) return BEVR_E_NULL;
  if (normalize && (!inv_norm_cam || !inv_norm_map)) return BEVR_E_NULL;
  if (n <= 0 || m <= 0 || E <= 0) return BEVR_E_SHAPE;
  if (((E & 3) == 0) && (!bevr_aligned16(cam) || !bevr_aligned16(map))) return BEVR_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (E + ECHUNK - 1) / ECHUNK;
  hipError_t e = hipMemsetAsync(D, 0, (size_t)n * m * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  if (n <= CORR_MAX_ROWS && m <= CORR_MAX_ROWS && (E & 3) == 0) {
    // training sizes: every embedding element read once, the row norms in the same pass
    if (normalize) {
      if ((e = hipMemsetAsync(inv_norm_cam, 0, n * sizeof(float), st)) != hipSuccess) return (int)e;
      if ((e = hipMemsetAsync(inv_norm_map, 0, m * sizeof(float), st)) != hipSuccess) return (int)e;
    }
    float* sqx = normalize ? inv_norm_cam : nullptr;
    float* sqy = normalize ? inv_norm_map : nullptr;
    const int fa = (n + 15) / 16, fb = (m + 15) / 16;
    const long long want = ((long long)E + 127) / 128;           // one 32-element step per wave at least
    const int grid = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
    // 8 waves per workgroup: twice the bytes in flight of 4 at the same number of final atomics (a streaming read needs
    // ~16 MB in flight to approach the HBM rate; BEVR_GRAM_WAVES=4 for A/B timing)
    static const int nw = getenv("BEVR_GRAM_WAVES") ? atoi(getenv("BEVR_GRAM_WAVES")) : 8;
#define BEVR_GRAM(FA_, FB_)                                                                                            \
  do {                                                                                                                 \
    if (nw == 4) hipLaunchKernelGGL((gram_mfma_kernel<FA_, FB_, 4>), dim3(grid), dim3(256), 0, st, cam, map, D, sqx, sqy, n, m, E); \
    else hipLaunchKernelGGL((gram_mfma_kernel<FA_, FB_, 8>), dim3(grid), dim3(512), 0, st, cam, map, D, sqx, sqy, n, m, E);         \
  } while (0)
    if (fa == 1 && fb == 1) BEVR_GRAM(1, 1);
    else if (fa <= 2 && fb <= 2) BEVR_GRAM(2, 2);
    else BEVR_GRAM(4, 4);
#undef BEVR_GRAM
    hipLaunchKernelGGL(finish_dist2_kernel, dim3(1), dim3(256), 0, st, D, inv_norm_cam, inv_norm_map, n, m, normalize);
    return (int)hipGetLastError();
  }
  if (normalize) {
    if ((e = hipMemsetAsync(inv_norm_cam, 0, n * sizeof(float), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(inv_norm_map, 0, m * sizeof(float), st)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(rownorm_kernel, dim3(chunks, n), dim3(256), 0, st, cam, inv_norm_cam, n, E);
    hipLaunchKernelGGL(rownorm_kernel, dim3(chunks, m), dim3(256), 0, st, map, inv_norm_map, m, E);
    hipLaunchKernelGGL(finish_norm_kernel, dim3((n + 63) / 64), dim3(64), 0, st, inv_norm_cam, n);
    hipLaunchKernelGGL(finish_norm_kernel, dim3((m + 63) / 64), dim3(64), 0, st, inv_norm_map, m);
  }
  hipLaunchKernelGGL(gram_kernel, dim3(chunks, (n + TB - 1) / TB, (m + TB - 1) / TB), dim3(256), 0, st, cam, map, D, n,
                     m, E);
  hipLaunchKernelGGL(finish_dist_kernel, dim3((n * m + 255) / 256), dim3(256), 0, st, D, inv_norm_cam, inv_norm_map, n,
                     m, normalize);
  return (int)hipGetLastError();
}

extern "C" int bevr_corr_bwd(const float* cam, const float* map, const float* D, const float* dD,
                             const float* inv_norm_cam, const float* inv_norm_map, float* dcam, float* dmap, int n,
                             int m, int E, int normalize, void* stream) {
  if (!cam || !map || !dD || !dcam || !dmap) return BEVR_E_NULL;
  if (normalize && (!inv_norm_cam || !inv_norm_map || !D)) return BEVR_E_NULL;
  if (n <= 0 || m <= 0 || E <= 0) return BEVR_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (E + 1023) / 1024;
  if (n <= CORR_MAX_ROWS && m <= CORR_MAX_ROWS && (E & 3) == 0 && bevr_aligned16(cam) && bevr_aligned16(map) &&
      bevr_aligned16(dcam) && bevr_aligned16(dmap)) {
    // cam == map (the retrieval losses correlate one embedding matrix with itself): dcam receives BOTH sides' sum and
    // dmap is not written -- bevrender_amd/ops.py hands the sum to autograd once
    if (cam == map && n == m)
      hipLaunchKernelGGL((corr_bwd_slice_kernel<true>), dim3(chunks), dim3(256), 0, st, cam, map, dD, D, inv_norm_cam,
                         inv_norm_map, dcam, dmap, n, m, E, normalize);
    else
      hipLaunchKernelGGL((corr_bwd_slice_kernel<false>), dim3(chunks), dim3(256), 0, st, cam, map, dD, D, inv_norm_cam,
                         inv_norm_map, dcam, dmap, n, m, E, normalize);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(corr_bwd_kernel, dim3(chunks, n), dim3(256), 0, st, cam, map, dD, D, inv_norm_cam, inv_norm_map,
                     dcam, n, m, E, 0, normalize);
  hipLaunchKernelGGL(corr_bwd_kernel, dim3(chunks, m), dim3(256), 0, st, map, cam, dD, D, inv_norm_map, inv_norm_cam,
                     dmap, m, n, E, 1, normalize);
  return (int)hipGetLastError();
}

extern "C" int bevr_recall_rank(const float* D, int32_t* rank, int n, void* stream) {
  if (!D || !rank) return BEVR_E_NULL;
  if (n <= 0) return BEVR_E_SHAPE;
  hipLaunchKernelGGL(recall_rank_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, D, rank, n);
  return (int)hipGetLastError();
}
