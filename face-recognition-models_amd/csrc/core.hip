// libfrx core: version, thread-local error text, device properties.
#include "frx_common.h"
#include <string.h>

namespace frx {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace frx

extern "C" int frx_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* frx_last_error(void) { return frx::g_err; }

// sizeof() of the structs that cross the boundary, in declaration order: bindings check their own layouts against it
extern "C" int frx_struct_sizes(int64_t sizes[8]) {
  FRX_CHECK_ARG(sizes != nullptr, "sizes is NULL");
  sizes[0] = (int64_t)sizeof(frx_head_desc);
  sizes[1] = (int64_t)sizeof(frx_conv_desc);
  sizes[2] = (int64_t)sizeof(frx_dgrad_fuse);
  sizes[3] = (int64_t)sizeof(frx_wgrad_job);
  sizes[4] = (int64_t)sizeof(frx_bn_tot);
  sizes[5] = sizes[6] = sizes[7] = 0;
  return FRX_OK;
}

extern "C" int frx_device_props(int device, int64_t props[8]) {
  FRX_CHECK_ARG(props != nullptr, "props is NULL");
  hipDeviceProp_t p;
  FRX_HIP(hipGetDeviceProperties(&p, device));
  props[0] = p.multiProcessorCount;
  props[1] = p.clockRate;
  props[2] = (int64_t)p.maxSharedMemoryPerMultiProcessor;
  props[3] = p.warpSize;
  props[4] = strstr(p.gcnArchName, "gfx950") != nullptr;
  props[5] = (int64_t)(p.totalGlobalMem >> 20);
  props[6] = p.memoryClockRate;
  props[7] = p.memoryBusWidth;
  return FRX_OK;
}
