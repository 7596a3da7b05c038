// Micro-benchmark: cost of LDS reads whose 64 lanes carry only TWO distinct addresses (one per 32-lane half) -- the
// per-(column, key) constants of the query-stationary attention kernels -- by access width.  16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ __launch_bounds__(1024) void k(int iters, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned a[16384];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 1024) a[i] = i;
  __syncthreads();
  unsigned acc = 0;
  const int wave = tid >> 6, lane = tid & 63;
  const int base = wave * 2048 + (lane >> 5) * 1024;   // two addresses per wave, 16-byte aligned
  for (int it = 0; it < iters; ++it) {
    if (W == 16) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(base + u * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].w; }
    } else if (W == 12) {
      u32x3 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b96 %0, %1" : "=v"(v[u]) : "v"(base + u * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].z; }
    } else if (W == 8) {
      u32x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(base + u * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].y; }
    } else if (W == 4) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u]) : "v"(base + u * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
    } else {   // W == 84: b64 + b32 of the same 16-byte record, as the kernels read their constants today
      u32x2 v[8]; unsigned c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(base + u * 16) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(c[u]) : "v"(base + u * 16) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u]), "+v"(c[u])); acc += v[u].x + v[u].y + c[u]; }
    }
  }
  out[blockIdx.x * 1024 + tid] = acc;
}
template <int W> float run() {
  static unsigned* out = nullptr;
  if (!out) hipMalloc(&out, 1024 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256;
  k<W><<<grid, 1024>>>(10, out);
  hipEventRecord(e0);
  k<W><<<grid, 1024>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 8 * 16) * 2.4;   // clk per record per CU
}
int main() {
  printf("two-address broadcast reads, clk per record (wave instruction) per CU\n");
  printf("ds_read_b32            %6.2f\n", run<4>());
  printf("ds_read_b64            %6.2f\n", run<8>());
  printf("ds_read_b96            %6.2f\n", run<12>());
  printf("ds_read_b128           %6.2f\n", run<16>());
  printf("ds_read_b64 + b32      %6.2f\n", run<84>());
  return 0;
}
