// Micro-benchmark: LDS instruction throughput per CU on gfx950 (16 waves per CU, conflict-free addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, unsigned* out, int stride) {
  __shared__ unsigned a[16384];
  __shared__ unsigned long long b[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 1024) a[i] = 0;
  for (int i = tid; i < 8192; i += 1024) b[i] = 0;
  __syncthreads();
  unsigned acc = 0;
  const int wave = tid >> 6, lane = tid & 63;
  int idx = (wave * 64 * stride + lane * stride) & 8191;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = (idx + u * 64) & 8191;
      if (MODE == 0) atomicAdd(&a[j], (unsigned)it);
      else if (MODE == 1) atomicAdd(&b[j], (unsigned long long)it);
      else if (MODE == 2) atomicAdd(reinterpret_cast<float*>(&a[j]), 1.0f);
      else if (MODE == 3) { a[j] = it; asm volatile("" ::: "memory"); }
      else if (MODE == 4) { acc += a[j]; asm volatile("" ::: "memory"); }
      else if (MODE == 5) { acc += (unsigned)b[j]; asm volatile("" ::: "memory"); }
      else if (MODE == 7) { const uint4 v = *reinterpret_cast<const uint4*>(&a[((idx >> 6) * 4 + u * 8 + (lane >> 5) * 16) & 16380]); acc += v.x + v.w; asm volatile("" ::: "memory"); }
      else if (MODE == 8) { const uint4 v = *reinterpret_cast<const uint4*>(&a[(j * 4) & 16380]); acc += v.x + v.w; asm volatile("" ::: "memory"); }
      else if (MODE == 9) { acc += a[((idx >> 6) + u * 8 + (lane >> 5) * 16) & 16383]; asm volatile("" ::: "memory"); }
      else if (MODE == 10) { const int q = (j * 2) & 8190; acc += (unsigned)b[q] + (unsigned)b[q + 1]; asm volatile("" ::: "memory"); }
      else if (MODE == 11) { typedef unsigned long long u64x2 __attribute__((ext_vector_type(2))); u64x2 v; const int q = ((j * 2) & 8190) * 8 + 65536; asm volatile("ds_read2_b64 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(q) : "memory"); acc += (unsigned)v.x + (unsigned)v.y; }
      else if (MODE == 6) { unsigned v = it; asm volatile("ds_add_u32 %0, %1" :: "v"(j * 4), "v"(v) : "memory"); }
      else if (MODE == 12) { float v = 1.0f; asm volatile("ds_add_f32 %0, %1" :: "v"(j * 4), "v"(v) : "memory"); }
      else if (MODE == 13) { unsigned long long v = it; asm volatile("ds_add_u64 %0, %1" :: "v"(65536 + j * 8), "v"(v) : "memory"); }
      else if (MODE == 14) { unsigned long long v; const int q = (j * 8 + 4) & 65535; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(q) : "memory"); acc += (unsigned)v + (unsigned)(v >> 32); }
      else if (MODE == 15) { float v = 1.0f; if ((lane & 31) == 31) asm volatile("ds_add_f32 %0, %1" :: "v"(j * 4), "v"(v) : "memory"); }
      else if (MODE == 16) { unsigned v = 0x3f803f80u; asm volatile("ds_pk_add_bf16 %0, %1" :: "v"(j * 4), "v"(v) : "memory"); }
      else if (MODE == 17) { unsigned long long v; const int q = (j * 8) & 65535; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(q) : "memory"); acc += (unsigned)v + (unsigned)(v >> 32); }
      else if (MODE == 18) { float v = 1.0f; asm volatile("ds_add_f32 %0, %1\n ds_add_f32 %0, %1 offset:256" :: "v"(j * 4), "v"(v) : "memory"); }
    }
  }
  if (MODE == 6 || MODE >= 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * 1024 + tid] = acc + a[tid] + (unsigned)b[tid];
}
template <int MODE> void run(const char* name, int stride) {
  unsigned* out; hipMalloc(&out, 1024 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256;
  k<MODE><<<grid, 1024>>>(10, out, stride);
  hipEventRecord(e0);
  k<MODE><<<grid, 1024>>>(iters, out, stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_cu = (double)iters * 8 * 16;   // wave instructions per CU
  printf("%-28s stride %d: %.3f ms, %.2f ns per wave-instr per CU (= %.1f clk @2.4GHz)\n", name, stride, ms,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
  hipFree(out);
}
int main() {
  for (int stride = 1; stride <= 2; ++stride) {
    run<0>("atomicAdd u32", stride);
    run<6>("ds_add_u32 asm noret", stride);
    run<1>("atomicAdd u64", stride);
    run<2>("atomicAdd f32", stride);
    run<3>("store b32", stride);
    run<4>("load b32", stride);
    run<5>("load b64", stride);
    run<7>("load b128 broadcast(2 addr)", stride);
    run<8>("load b128 per-lane", stride);
    run<9>("load b32 broadcast(2 addr)", stride);
    run<10>("load 2 x b64 (compiler)", stride);
    run<11>("ds_read2_b64 asm", stride);
    run<12>("ds_add_f32 asm noret", stride);
    run<13>("ds_add_u64 asm noret", stride);
    run<17>("ds_read_b64 asm aligned", stride);
    run<14>("ds_read_b64 asm addr%8==4", stride);
    run<15>("ds_add_f32 2 lanes active", stride);
    run<16>("ds_pk_add_bf16 asm noret", stride);
    run<18>("2 x ds_add_f32 (per pair)", stride);
  }
  return 0;
}
