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
