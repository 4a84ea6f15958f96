#include "reduce.hpp"

namespace mml {

__global__ __launch_bounds__(256) void slab_reduce_kernel(const ReduceLaunch R) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R.total; i += stride) {
    int si = 0;
    while (si + 1 < R.n && i >= R.seg[si + 1].start) ++si;
    const ReduceSeg& g = R.seg[si];
    const int64_t j = i - g.start;
    float s = 0.f;
    for (int k = 0; k < g.S; ++k) s += g.slab[(int64_t)k * g.sstride + j];
    const int64_t r = j / g.cols, c = j - r * g.cols;
    float* dst = g.out + r * g.ldo + c;
    if (g.accumulate) s += *dst;
    *dst = s;
  }
}

int launch_slab_reduce(const ReduceLaunch& R, hipStream_t st, const char* who) {
  if (R.total <= 0) return MML_OK;
  int64_t rb = cdiv(R.total, 256);
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, R);
  return check_launch(who);
}

}  // namespace mml
