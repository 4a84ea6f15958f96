// How fast can ONE wave per SIMD issue v_mfma_f32_32x32x16_f16 -- against two waves per SIMD?  (round 4: the panel
// kernel's pure MFMA loop ran at ~57 cycles per instruction with one wave per SIMD.)
// build: hipcc --offload-arch=gfx950 -O3 mfma_occ.hip -o mfma_occ ; run: ./mfma_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int CHAIN>
__global__ void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.02f * (threadIdx.x - e)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAIN; ++c)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int CHAIN>
void run(const char* name, int threads, int blocks, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, CHAIN>), dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) {
      const double mf = (double)iters * NACC * CHAIN;  // MFMAs per wave
      printf("%-44s %7.3f ms  %6.1f ns per MFMA and wave  -> %6.1f TFLOP/s\n", name, ms, ms * 1e6 / mf,
             mf * (threads / 64) * blocks * 32.0 * 32 * 16 * 2 / (ms * 1e-3) / 1e12);
    }
  }
}
int main() {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  run<4, 3>("1 wave/SIMD (256 thr x 256 WG), 4 acc x 3", 256, 256, d);
  run<4, 3>("2 waves/SIMD (512 thr x 256 WG), 4 acc x 3", 512, 256, d);
  run<4, 3>("2 waves/SIMD (256 thr x 512 WG), 4 acc x 3", 256, 512, d);
  run<12, 1>("1 wave/SIMD, 12 independent acc", 256, 256, d);
  run<1, 12>("1 wave/SIMD, one chain", 256, 256, d);
  run<4, 3>("4 waves/SIMD (1024 thr x 256 WG)", 1024, 256, d);
  return 0;
}
