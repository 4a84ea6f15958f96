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
__device__ __forceinline__ void amax_acc(uint32_t& am, float v) {
  const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
  am = b > am ? b : am;
}
__device__ __forceinline__ void amax_acc(uint32_t& am, const float4& v) {
  amax_acc(am, v.x); amax_acc(am, v.y); amax_acc(am, v.z); amax_acc(am, v.w);
}
// Call with the whole wave converged.  Wave maximum, then at most ONE atomic per wave -- and none when the slot word
// already holds a value at least as large (slots only ever grow during a launch, so a stale read can only cost an
// unnecessary atomic; thousands of waves hammering one address serialise at the memory side otherwise).
__device__ __forceinline__ void amax_flush(uint32_t am, uint32_t* slot) {
  if (!slot) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)am, o, 64);
    am = t > am ? t : am;
  }
  if ((threadIdx.x & 63) == 0 && am) {
    uint32_t* p = slot + (blockIdx.x & (MML_AMAX_WORDS - 1));
    if (__atomic_load_n(p, __ATOMIC_RELAXED) < am) atomicMax(p, am);
  }
}
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
