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
