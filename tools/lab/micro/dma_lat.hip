// Microbenchmark: latency of one LDS-DMA wave-instruction (issue -> s_waitcnt vmcnt(0)) for L2-resident and for
// streamed (HBM / MALL) sources, with 1 and 8 waves per CU issuing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int DEPTH>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t window, int iters, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + ((size_t)blockIdx.x * 8 + wave) * window;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const char* g = base + ((size_t)(it * DEPTH + d) * 1024) % window + lane * 16;
      __builtin_amdgcn_global_load_lds((const float*)g, sm + (wave * DEPTH + d) * 256, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (sm[tid] == 12345.f) out[0] = 1.f;
}
template <int DEPTH>
void run(const char* src, size_t window, int threads, const char* what, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  CK(hipFuncSetAttribute((const void*)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<DEPTH>), dim3(256), dim3(threads), 8 * DEPTH * 1024, 0, src, window, 10, out, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<DEPTH>), dim3(256), dim3(threads), 8 * DEPTH * 1024, 0, src, window, iters, out, cyc);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-28s depth %d, %d waves/CU: %6.0f ns per round  (%5.1f GB/s per CU)\n", what, DEPTH, threads / 64, ms * 1e6 / iters,
         (double)(threads / 64) * DEPTH * 1024 * iters / ms / 1e6);
}
int main() {
  const size_t total = 2048u << 20;
  char* src; float* out; unsigned long long* cyc;
  CK(hipMalloc(&src, total)); CK(hipMemset(src, 0, total)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, 8));
  // L2-resident: 8 KiB window per wave (re-read); streamed: 1 MiB window per wave (2 GiB in all: misses every cache)
  for (int threads : {64, 512}) {
    run<1>(src, 8192, threads, "L2-resident window", out, cyc);
    run<4>(src, 8192, threads, "L2-resident window", out, cyc);
    run<1>(src, 1u << 20, threads, "streamed (HBM)", out, cyc);
    run<4>(src, 1u << 20, threads, "streamed (HBM)", out, cyc);
  }
  return 0;
}
