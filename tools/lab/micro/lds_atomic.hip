// Microbenchmark: LDS accumulate throughput on gfx950.  256-thread workgroups, 4 per CU, each lane does ITERS
// accumulate operations into a 32 KiB LDS array at (a) conflict-free consecutive, (b) random, (c) same-address slots.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int N = 8192;   // floats of LDS per workgroup
constexpr int ITERS = 256;

template <int MODE, int PAT>
__global__ __launch_bounds__(256) void k(float* out, const int* rnd) {
  __shared__ float acc[N];
  for (int i = threadIdx.x; i < N; i += 256) acc[i] = 0.f;
  __syncthreads();
  unsigned a = PAT == 0 ? threadIdx.x : (PAT == 1 ? rnd[threadIdx.x + blockIdx.x * 256] : (threadIdx.x >> 3));
  float v = 1.0f + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    const unsigned idx = (PAT == 2 ? a : (a + it * 257u)) & (N - 1);
    if (MODE == 0) atomicAdd(&acc[idx], v);                                  // ds_add_f32
    else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(acc) + idx, 3u);  // ds_add_u32
    else if (MODE == 2) acc[idx] += v;                                       // ds_read + ds_write (racy)
    else if (MODE == 3) acc[idx] = v;                                        // ds_write only
    else if (MODE == 4) { float o = __hip_atomic_fetch_add(&acc[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); v += o * 1e-30f; }  // rtn
    else if (MODE == 5) { unsigned o = atomicAdd(reinterpret_cast<unsigned*>(acc) + idx, 3u); v += o * 1e-30f; }  // ds_add_rtn_u32
    else if (MODE == 6) { int o = atomicCAS(reinterpret_cast<int*>(acc) + idx, 0, (int)a | 1); v += o * 1e-30f; }   // ds_cmpst_rtn_b32
    else if (MODE == 7) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (idx >> 1), 0x100000003ull);  // ds_add_u64
    else if (MODE == 8) { unsigned long long o = atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (idx >> 1), 0x100000003ull); v += (unsigned)o * 1e-30f; }
    else if (MODE == 9) atomicMax(reinterpret_cast<int*>(acc) + idx, (int)a);  // ds_max_i32
    if (PAT == 1) a = a * 1664525u + 1013904223u;
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += acc[i];
  if (s == 12345.f) out[0] = s;
}

template <int MODE, int PAT>
void run(const char* name, float* out, int* rnd) {
  const int blocks = 256 * 4 * 4;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(256), 0, 0, out, rnd);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(256), 0, 0, out, rnd);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  const double wave_instr_per_cu = (double)blocks * 4 * ITERS / 256;
  printf("%-28s %8.1f us   %.1f cycles/wave-instr/CU (2.4GHz)   %.2f Glane-ops/s\n", name, ms * 1e3,
         ms * 1e-3 * 2.4e9 / wave_instr_per_cu, (double)blocks * 256 * ITERS / (ms * 1e-3) / 1e9);
}

int main() {
  float* out; int* rnd;
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&rnd, 256 * 4 * 4 * 256 * 4));
  int* h = (int*)malloc(256 * 4 * 4 * 256 * 4);
  for (int i = 0; i < 256 * 4 * 4 * 256; ++i) h[i] = rand();
  CK(hipMemcpy(rnd, h, 256 * 4 * 4 * 256 * 4, hipMemcpyHostToDevice));
  run<0, 0>("ds_add_f32 consecutive", out, rnd);
  run<0, 1>("ds_add_f32 random", out, rnd);
  run<0, 2>("ds_add_f32 8 lanes/addr", out, rnd);
  run<1, 0>("ds_add_u32 consecutive", out, rnd);
  run<1, 1>("ds_add_u32 random", out, rnd);
  run<2, 0>("read+write consecutive", out, rnd);
  run<2, 1>("read+write random", out, rnd);
  run<3, 1>("write random", out, rnd);
  run<4, 0>("ds_add_rtn_f32 consecutive", out, rnd);
  run<4, 1>("ds_add_rtn_f32 random", out, rnd);
  run<5, 1>("ds_add_rtn_u32 random", out, rnd);
  run<6, 1>("ds_cmpst_rtn_b32 random", out, rnd);
  run<6, 2>("ds_cmpst_rtn_b32 8 lanes/addr", out, rnd);
  run<7, 0>("ds_add_u64 consecutive", out, rnd);
  run<7, 1>("ds_add_u64 random", out, rnd);
  run<7, 2>("ds_add_u64 8 lanes/addr", out, rnd);
  run<8, 1>("ds_add_rtn_u64 random", out, rnd);
  run<9, 1>("ds_max_i32 random", out, rnd);
  run<1, 2>("ds_add_u32 8 lanes/addr", out, rnd);
  return 0;
}
