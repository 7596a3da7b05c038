// Micro-benchmark: cost of ds_read_b128 / ds_read_b64 per wave instruction as a function of the lane -> address map
// (16 waves per CU issuing back to back, gfx950).  Which fragment layouts are conflict-free?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ int lane_addr(int pat, int lane) {
  const int lq = lane & 31, hi = lane >> 5;
  switch (pat) {
    case 0: return lane * 16;                                   // contiguous
    case 1: return lq * 80 + hi * 16;                           // K rows, 64 B + 16 pad, chunk hi
    case 2: return lq * 64 + hi * 16;                           // K rows unpadded
    case 3: return lq * 144 + hi * 16;                          // V^T rows, 128 B + 16 pad
    case 4: return lq * 32 + hi * 16;                           // two lanes halves interleaved: row = 32 B
    case 5: return (lane & 7) * 16 + (lane >> 3) * 128 + 0;     // contiguous (same as 0)
    case 6: return lq * 16 + hi * 512 + hi * 64;                // halves 576 B apart
    case 7: return lq * 16 + hi * 512;                          // halves 512 B apart
    case 8: return ((lane & 3) * 16) + ((lane >> 2) & 1) * 64 + (lane >> 3) * 128;   // contiguous again (sanity)
    case 9: return lq * 80 + hi * 32;                           // K rows padded, chunks 32 B apart
    case 10: return lq * 48 + hi * 16;                          // 32 B rows + 16 pad
    case 11: return lq * 272 + hi * 16;                         // 256 B + 16 pad
    case 12: return (lane ^ ((lane >> 3) & 1)) * 16;            // swap neighbours in odd groups
    case 13: return lane * 16 + (lane >> 3) * 16;               // +16 B skew per 8 lanes
    case 14: return lane * 16 + (lane >> 4) * 64;               // +64 B skew per 16 lanes
    case 15: return lane * 16 + (lane >> 5) * 64;               // +64 B between halves
    case 16: return lq * 64 + hi * 16 + ((lq >> 1) & 1) * 32;   // unpadded rows with chunk swizzle
    case 17: return lq * 64 + ((hi ^ (lq & 1)) * 16) + ((lq >> 1) & 1) * 32;
  }
  return 0;
}

template <int W>
__global__ __launch_bounds__(1024) void k(int iters, unsigned* out, int pat) {
  __shared__ __attribute__((aligned(16))) unsigned a[16384];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 1024) a[i] = i;
  __syncthreads();
  unsigned acc = 0;
  const int wave = tid >> 6, lane = tid & 63;
  const int base = (lane_addr(pat, lane) + wave * 1024) & 0x7fff;   // within the lower 32 KB
  for (int it = 0; it < iters; ++it) {
    // 8 independent reads in flight, one wait: the LDS pipe, not the latency, is what is measured
    if (W == 16) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(base + u * 2048) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].w; }
    } else if (W == 8) {
      u32x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(base + u * 2048) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].y; }
    } else {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u]) : "v"(base + u * 2048) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
    }
  }
  out[blockIdx.x * 1024 + tid] = acc;
}
template <int W> float run(int pat) {
  static unsigned* out = nullptr;
  if (!out) hipMalloc(&out, 1024 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256;
  k<W><<<grid, 1024>>>(10, out, pat);
  hipEventRecord(e0);
  k<W><<<grid, 1024>>>(iters, out, pat);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 8 * 16) * 2.4;   // clk per wave instruction per CU
}
int main() {
  const char* names[] = {"contiguous 16 B/lane", "row 80 B, chunk hi*16", "row 64 B, chunk hi*16", "row 144 B, chunk hi*16",
                         "row 32 B, chunk hi*16", "contiguous (8-lane groups)", "lq*16, halves +576", "lq*16, halves +512",
                         "contiguous (4-lane groups)", "row 80 B, chunk hi*32", "row 48 B, chunk hi*16", "row 272 B, chunk hi*16",
                         "swap in odd 8-groups", "+16 B skew per 8 lanes", "+64 B skew per 16 lanes", "+64 B between halves",
                         "row 64 B swizzle a", "row 64 B swizzle b"};
  printf("%-30s %8s %8s %8s   (clk per wave instruction per CU)\n", "lane -> address", "b128", "b64", "b32");
  for (int p = 0; p < 18; ++p) printf("%-30s %8.1f %8.1f %8.1f\n", names[p], run<16>(p), run<8>(p), run<4>(p));
  return 0;
}
