// Shared helpers for the MI355X (gfx950) kernels behind include/mmlrec.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mmlrec.h"

namespace mml {

constexpr int kWave = 64;  // CDNA wavefront width

void set_error(const char* fmt, ...);

inline hipStream_t to_stream(mml_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Check a launch without synchronising (capturable).
int check_launch(const char* what);

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// ---- operand magnitudes (include/mmlrec.h): per-lane running maximum of |x| bit patterns, published at kernel end ----
__device__ __forceinline__ void amax_acc(float& am, float v) { am = fmaxf(am, fabsf(v)); }  // one v_max_f32 (|x| modifier)
__device__ __forceinline__ void amax_acc(float& am, const float4& v) {
  am = fmaxf(fmaxf(am, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
}
// End of a kernel, every thread of the workgroup converged (<= 16 waves): wave maxima -> LDS -> ONE atomic per workgroup
// on the slot word the workgroup number picks.  Kernels that use it run a few thousand workgroups at most; same-address
// atomics serialise at the memory side (~11 ns each), and a device-coherent "is it larger already?" read per wave is no
// cheaper than the atomic it would save (measured on the gather: 15 000 workgroups -> +0.1 ms either way, so the
// gather's output is measured by the stand-alone pass instead).
// (A NaN never registers in the running maximum; it reaches the consumer as a NaN whatever the scale.)
template <bool LOOK = false>
__device__ __forceinline__ void amax_flush(float amf, uint32_t* slot) {
  if (!slot) return;  // (uniform)
  __shared__ float amax_wg[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amf = fmaxf(amf, __shfl_xor(amf, o, 64));
  if ((threadIdx.x & 63) == 0) amax_wg[threadIdx.x >> 6] = amf;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    float m = amax_wg[0];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, amax_wg[w]);
    const uint32_t am = __float_as_uint(m);
    // LOOK before the atomic: the words only ever rise, so a workgroup whose maximum is not above what it reads has
    // nothing to add (a stale-low read costs one needless atomic, never a wrong slot) -- the atomics of a launch all
    // land on the slot's one 32-byte line and serialise there (~8 ns each).  Only for kernels whose workgroups live
    // long (the stand-alone magnitude pass: 35 -> 20 us at 2 048 workgroups): in the row kernels the extra memory round
    // trip at the end of every short-lived workgroup cost more than the atomics it saved (gate backward 96 -> 121 us).
    uint32_t* const w = slot + (blockIdx.x & (MML_AMAX_WORDS - 1));
    if (LOOK) {
      if (am && am > __builtin_nontemporal_load(w)) atomicMax(w, am);
    } else if (am) {
      atomicMax(w, am);
    }
  }
  __syncthreads();  // (the array may be reused by the next flush of the same kernel)
}
#endif

}  // namespace mml

namespace mml {
#if defined(__HIPCC__)
// torch.clamp(x, min = -100) of F.binary_cross_entropy's log terms: a NaN stays a NaN (fmaxf alone returns the other
// operand, and a diverged model would keep reporting a finite loss)
__device__ __forceinline__ float bce_log_clamp(float x) { return x != x ? x : fmaxf(x, -100.f); }
#endif
}  // namespace mml

// hipGetLastError() is sticky per host thread: an error left behind by an unrelated earlier HIP call (e.g. a device
// probe before the runtime was initialised) must not be blamed on our launch, so clear it first.
#define MML_LAUNCH(...)          \
  do {                           \
    (void)hipGetLastError();     \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

#define MML_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) {                              \
      mml::set_error(__VA_ARGS__);              \
      return MML_ERR_ARG;                       \
    }                                           \
  } while (0)
