#include "common.hpp"

#include <stdarg.h>

namespace mml {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MML_ERR_HIP;
  }
  return MML_OK;
}

}  // namespace mml

extern "C" int mml_version(void) { return 100; }

extern "C" const char* mml_last_error(void) { return mml::g_err; }

extern "C" int mml_device_caps(int device, int64_t* out6) {
  MML_REQUIRE(out6 != nullptr, "mml_device_caps: out6 is null");
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    mml::set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return MML_ERR_HIP;
  }
  out6[0] = p.multiProcessorCount;
  out6[1] = (int64_t)p.maxSharedMemoryPerMultiProcessor;
  out6[2] = (int64_t)p.totalGlobalMem;
  out6[3] = p.warpSize;
  out6[4] = p.clockRate;
  int arch = 0;
  const char* g = strstr(p.gcnArchName, "gfx");
  if (g) sscanf(g + 3, "%d", &arch);
  out6[5] = arch;
  return MML_OK;
}

extern "C" int mml_stream_create_cu_range(int device, int cu_lo, int cu_hi, mml_stream_t* stream_out) {
  MML_REQUIRE(stream_out != nullptr, "mml_stream_create_cu_range: stream_out is null");
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    mml::set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return MML_ERR_HIP;
  }
  const int ncu = p.multiProcessorCount;
  MML_REQUIRE(0 <= cu_lo && cu_lo < cu_hi && cu_hi <= ncu, "mml_stream_create_cu_range: need 0 <= lo < hi <= CU count");
  uint32_t mask[32] = {0};
  MML_REQUIRE(ncu <= 32 * 32, "mml_stream_create_cu_range: more than 1024 compute units");
  for (int i = cu_lo; i < cu_hi; ++i) mask[i >> 5] |= 1u << (i & 31);
  int prev = 0;
  hipGetDevice(&prev);
  if (prev != device) hipSetDevice(device);
  hipStream_t s = nullptr;
  e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((ncu + 31) / 32), mask);
  if (prev != device) hipSetDevice(prev);
  if (e != hipSuccess) {
    mml::set_error("hipExtStreamCreateWithCUMask([%d, %d)): %s", cu_lo, cu_hi, hipGetErrorString(e));
    return MML_ERR_HIP;
  }
  *stream_out = (mml_stream_t)s;
  return MML_OK;
}

extern "C" int mml_stream_destroy(mml_stream_t stream) {
  MML_REQUIRE(stream != nullptr, "mml_stream_destroy: null stream");
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    mml::set_error("hipStreamDestroy: %s", hipGetErrorString(e));
    return MML_ERR_HIP;
  }
  return MML_OK;
}
