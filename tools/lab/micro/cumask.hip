// Microbenchmark: HBM streaming rate of a CU-masked stream (hipExtStreamCreateWithCUMask) as a function of how many
// compute units the mask holds and which ones (contiguous bits vs every 8th bit), plus the XCC_ID the waves report.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// Adam-like stream: 4 reads + 3 writes of 16 B per lane and iteration, U independent iterations in flight
template <int U>
__global__ __launch_bounds__(256) void stream_k(const float4* __restrict__ a, const float4* __restrict__ b,
                                                float4* __restrict__ c, float4* __restrict__ d, long n, unsigned* xcc) {
  long i = (long)blockIdx.x * 256 * U + threadIdx.x;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    atomicAdd(&xcc[id & 15], 1u);
  }
  float4 va[U], vb[U], vc[U], vd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { long j = i + u * 256; if (j < n) { va[u] = a[j]; vb[u] = b[j]; vc[u] = c[j]; vd[u] = d[j]; } }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    long j = i + u * 256;
    if (j < n) {
      float4 x = va[u], y = vb[u], z = vc[u], w = vd[u];
      z.x = 0.9f * z.x + 0.1f * y.x; z.y = 0.9f * z.y + 0.1f * y.y; z.z = 0.9f * z.z + 0.1f * y.z; z.w = 0.9f * z.w + 0.1f * y.w;
      w.x = 0.99f * w.x + 0.01f * y.x * y.x; w.y = 0.99f * w.y + 0.01f * y.y * y.y; w.z = 0.99f * w.z + 0.01f * y.z * y.z; w.w = 0.99f * w.w + 0.01f * y.w * y.w;
      x.x -= 0.001f * z.x * __frsqrt_rn(w.x + 1e-8f); x.y -= 0.001f * z.y * __frsqrt_rn(w.y + 1e-8f);
      x.z -= 0.001f * z.z * __frsqrt_rn(w.z + 1e-8f); x.w -= 0.001f * z.w * __frsqrt_rn(w.w + 1e-8f);
      const_cast<float4*>(a)[j] = x; c[j] = z; d[j] = w;
    }
  }
}
template <int U>
float run(hipStream_t s, float4* a, float4* b, float4* c, float4* d, long n, unsigned* xcc) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int grid = (int)((n + 256 * U - 1) / (256 * U));
  hipLaunchKernelGGL(stream_k<U>, dim3(grid), dim3(256), 0, s, a, b, c, d, n, xcc);
  CK(hipStreamSynchronize(s));
  CK(hipMemsetAsync(xcc, 0, 64, s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(stream_k<U>, dim3(grid), dim3(256), 0, s, a, b, c, d, n, xcc);
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}
int main() {
  const long n = 24L << 20;  // float4 per array: 384 MiB each, 4 arrays; 7 x 16 B per element of traffic
  float4 *a, *b, *c, *d; unsigned* xcc;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&d, n * 16)); CK(hipMalloc(&xcc, 64));
  CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c, 0, n * 16)); CK(hipMemset(d, 0, n * 16));
  struct { const char* name; int lo, hi, stride; } cases[] = {
      {"all 256", 0, 256, 1}, {"[0,128)", 0, 128, 1}, {"[0,64)", 0, 64, 1}, {"[0,32)", 0, 32, 1}, {"[128,256)", 128, 256, 1},
      {"every 2nd (128)", 0, 256, 2}, {"every 8th (32)", 0, 256, 8}, {"every 4th (64)", 0, 256, 4}, {"[0,192)", 0, 192, 1}, {"[0,96)", 0, 96, 1}};
  for (auto& cs : cases) {
    uint32_t mask[8] = {0};
    int cnt = 0;
    for (int i = cs.lo; i < cs.hi; i += cs.stride) { mask[i >> 5] |= 1u << (i & 31); ++cnt; }
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    float t1 = run<1>(s, a, b, c, d, n, xcc);
    float t4 = run<4>(s, a, b, c, d, n, xcc);
    unsigned h[16]; CK(hipMemcpy(h, xcc, 64, hipMemcpyDeviceToHost));
    double gb = n * 16.0 * 7 / 1e9;
    printf("%-18s %3d CUs : U=1 %7.3f ms %7.1f GB/s | U=4 %7.3f ms %7.1f GB/s | workgroups per XCC:", cs.name, cnt, t1, gb / t1 * 1e3, t4, gb / t4 * 1e3);
    for (int x = 0; x < 8; ++x) printf(" %u", h[x]);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}
