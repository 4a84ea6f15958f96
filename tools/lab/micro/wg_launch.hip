// Microbenchmark: cost of dispatching many workgroups that do (almost) nothing, as a function of their LDS footprint.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k(float* out, int n) {
  extern __shared__ float sm[];
  if (n == 12345) { sm[threadIdx.x] = 1.f; __syncthreads(); out[0] = sm[255 - threadIdx.x]; }
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int ldss[] = {0, 16, 38, 76, 150};
  const int grids[] = {512, 2400, 3840, 16384};
  for (int lds : ldss) for (int grid : grids) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds * 1024, 0, out, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds * 1024, 0, out, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("lds %3d KiB  grid %5d : %7.1f us per launch  (%.1f ns per workgroup)\n", lds, grid, ms * 100, ms * 1e5 / grid);
  }
  return 0;
}
