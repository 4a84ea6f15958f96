// Microbenchmark: global -> LDS bandwidth of one CU-filling workgroup set, LDS-DMA (global_load_lds_dwordx4) against
// register-staged (global_load_dwordx4 + ds_write_b128), for lane-contiguous and row-scattered sources (L2-resident).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
// MODE 0: DMA, 1: register-staged.  Each wave moves NI x 1 KiB per iteration; PITCH = bytes between the 16-byte pieces of
// consecutive lanes (16 = contiguous, larger = one piece per row).
template <int MODE, int NI>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t span, int iters, int pitch, float* out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + ((size_t)blockIdx.x * 8 + wave) * 65536 % span;
  float accum = 0.f;
  for (int it = 0; it < iters; ++it) {
    f4 r[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const char* g = base + (size_t)((it * NI + j) % 32) * 2048 + (size_t)lane * pitch;
      float* l = sm + (wave * NI + j) * 256;
      if (MODE == 0) __builtin_amdgcn_global_load_lds((const float*)g, l, 16, 0, 0);
      else r[j] = *(const f4*)g;
    }
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < NI; ++j) *(f4*)(sm + (wave * NI + j) * 256 + lane * 4) = r[j];
    }
    if ((it & 3) == 3) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      accum += sm[(tid * 7) & 1023];
      __syncthreads();
    }
  }
  if (accum == 12345.f) out[0] = accum;
}
template <int MODE, int NI>
void run(const char* src, size_t span, int pitch, float* out, const char* name) {
  const int iters = 2000;
  CK(hipFuncSetAttribute((const void*)k<MODE, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE, NI>), dim3(256), dim3(512), 8 * NI * 1024, 0, src, span, 10, pitch, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<MODE, NI>), dim3(256), dim3(512), 8 * NI * 1024, 0, src, span, iters, pitch, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double bytes = 256.0 * 8 * NI * 1024 * iters;
  printf("%-34s pitch %4d : %7.1f GB/s per CU  (%.2f TB/s)\n", name, pitch, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
}
int main() {
  const size_t span = 64u << 20;
  char* src; float* out;
  CK(hipMalloc(&src, span + (1 << 20))); CK(hipMemset(src, 0, span + (1 << 20))); CK(hipMalloc(&out, 4));
  for (int pitch : {16, 48, 480, 1440}) {
    run<0, 5>(src, span, pitch, out, "LDS-DMA dwordx4, 5 per wave");
    run<1, 5>(src, span, pitch, out, "global_load_dwordx4 + ds_write");
  }
  return 0;
}
