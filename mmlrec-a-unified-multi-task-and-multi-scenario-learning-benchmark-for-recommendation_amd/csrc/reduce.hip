#include "reduce.hpp"

namespace mml {

// Many outputs, few partials (wgrad): one thread per output element, coalesced across outputs.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const ReduceLaunch R) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R.total; i += stride) {
    int si = 0;
    while (si + 1 < R.n && i >= R.seg[si + 1].start) ++si;
    const ReduceSeg& g = R.seg[si];
    const int64_t j = i - g.start;
    float s = 0.f;
    const float* src = g.slab + j;
    int k = 0;
    // eight independent loads in flight, then the adds in the fixed order k = 0, 1, 2, ... (bitwise reproducible)
    for (; k + 8 <= g.S; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u) * g.sstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < g.S; ++k) s += src[(int64_t)k * g.sstride];
    const int64_t r = j / g.cols, c = j - r * g.cols;
    float* dst = g.out + r * g.ldo + c;
    if (g.accumulate) s += *dst;
    *dst = s;
  }
}

// Few outputs, many partials (row kernels: one partial per workgroup): one WAVEFRONT per output element.  Lane l sums
// partials l, l+64, ... (64 independent load chains instead of one), then a fixed butterfly combines the lanes:
// the summation order depends only on S, so the result is bitwise reproducible.
__global__ __launch_bounds__(256) void slab_reduce_wave_kernel(const ReduceLaunch R) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < R.total; i += nwaves) {
    int si = 0;
    while (si + 1 < R.n && i >= R.seg[si + 1].start) ++si;
    const ReduceSeg& g = R.seg[si];
    const int64_t j = i - g.start;
    float s = 0.f;
    const float* src = g.slab + j;
    int k = lane;
    // eight loads in flight per lane, the adds in the order k = lane, lane + 64, ... (round 5: a lane's partials were
    // fetched one dependent load at a time -- 16 round trips for the 1 024 partials of a gate kernel's grid)
    for (; k + 7 * 64 < g.S; k += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + 64 * u) * g.sstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < g.S; k += 64) s += src[(int64_t)k * g.sstride];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      const int64_t r = j / g.cols, c = j - r * g.cols;
      float* dst = g.out + r * g.ldo + c;
      if (g.accumulate) s += *dst;
      *dst = s;
    }
  }
}

// In between (a small layer's weight gradient cut into ~200 batch chunks: 16 k outputs x 192 partials): 64 consecutive
// outputs per workgroup, its 16 waves deal the partials (wave w takes k = w, w + 16, ...: every load is 256 contiguous
// bytes), partial sums meet in LDS and are added in wave order -- fixed order, bitwise reproducible.  The
// one-wave-per-output form reads one 4-byte piece per cache line here: 20 us for 12.7 MB (tower layers of AE-30 at
// B = 65 536); this one reads whole lines.
__global__ __launch_bounds__(1024) void slab_reduce_block_kernel(const ReduceLaunch R) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < R.total; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + lane;
    float s = 0.f;
    int si = 0;
    int64_t j = 0;
    if (i < R.total) {
      while (si + 1 < R.n && i >= R.seg[si + 1].start) ++si;
      const ReduceSeg& g = R.seg[si];
      j = i - g.start;
      const float* src = g.slab + j;
      int k = w;
      for (; k + 7 * 16 < g.S; k += 8 * 16) {  // eight loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + 16 * u) * g.sstride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < g.S; k += 16) s += src[(int64_t)k * g.sstride];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < R.total) {
      float t = part[0][lane];
#pragma unroll
      for (int q = 1; q < 16; ++q) t += part[q][lane];
      const ReduceSeg& g = R.seg[si];
      const int64_t r = j / g.cols, c = j - r * g.cols;
      float* dst = g.out + r * g.ldo + c;
      if (g.accumulate) t += *dst;
      *dst = t;
    }
    __syncthreads();
  }
}

static thread_local ReduceLaunch* g_collect = nullptr;
void reduce_collect(ReduceLaunch* into) { g_collect = into; }

int launch_slab_reduce(const ReduceLaunch& R, hipStream_t st, const char* who) {
  if (R.total <= 0) return MML_OK;
  if (g_collect) {
    ReduceLaunch& C = *g_collect;
    if (C.n + R.n > MAX_REDUCE_SEGS) {
      set_error("%s: more than %d reduction segments in one batch", who, MAX_REDUCE_SEGS);
      return MML_ERR_ARG;
    }
    for (int i = 0; i < R.n; ++i) {
      C.seg[C.n] = R.seg[i];
      C.seg[C.n].start += C.total;
      ++C.n;
    }
    C.total += R.total;
    return MML_OK;
  }
  int maxS = 0;
  for (int i = 0; i < R.n; ++i)
    if (R.seg[i].S > maxS) maxS = R.seg[i].S;
  if (R.total >= 2048 && R.total <= 32768 && maxS >= 64) {  // (above: one thread per output is faster, measured)
    int64_t rb = cdiv(R.total, 64);
    if (rb > 2048) rb = 2048;
    MML_LAUNCH(slab_reduce_block_kernel, dim3((unsigned)rb), dim3(1024), 0, st, R);
  } else if (R.total <= 32768 && maxS >= 64) {
    int64_t rb = cdiv(R.total, 4);  // 4 waves per workgroup
    if (rb > 4096) rb = 4096;
    MML_LAUNCH(slab_reduce_wave_kernel, dim3((unsigned)rb), dim3(256), 0, st, R);
  } else {
    int64_t rb = cdiv(R.total, 256);
    if (rb > 2048) rb = 2048;
    MML_LAUNCH(slab_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, R);
  }
  return check_launch(who);
}

}  // namespace mml
