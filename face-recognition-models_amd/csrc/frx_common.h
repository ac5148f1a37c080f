// Internal helpers shared by the libfrx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/frx.h"

namespace frx {

void set_error(const char* fmt, ...);

#define FRX_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      frx::set_error(__VA_ARGS__);               \
      return FRX_ERR_ARG;                        \
    }                                            \
  } while (0)

#define FRX_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      frx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return FRX_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

// after a kernel launch: catches launch-configuration errors without synchronising
#define FRX_LAUNCH_CHECK() FRX_HIP(hipGetLastError())

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    target = dev;
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != target) (void)hipSetDevice(prev);
  }
  int target = -1;
};

#define FRX_ENTER(device)                                     \
  frx::DeviceGuard guard_(device);                            \
  if (!guard_.ok) {                                           \
    frx::set_error("cannot select HIP device %d", device);    \
    return FRX_ERR_HIP;                                       \
  }

static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline bool frx_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ---- device-side numeric helpers -------------------------------------------------
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide reductions for 256-thread blocks (4 waves); `sh` needs >= 4 floats
__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max256(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

}  // namespace frx
