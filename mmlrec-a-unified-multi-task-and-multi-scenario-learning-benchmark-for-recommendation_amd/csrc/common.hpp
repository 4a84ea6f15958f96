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
