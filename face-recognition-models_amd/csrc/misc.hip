// Optimiser, weight-layout preparation, input staging, casts.  HBM-bound helpers.
// Reference call sites: optim.SGD(momentum 0.9, wd 5e-4) model_utils.py:557,186;
// ToTensor + Normalize(0.5, 0.5) model_utils.py:539-547; .to(device) :173.
#include "conv_kernels.h"

namespace frx {

// torch.optim.SGD: d = g*gscale + wd*p;  buf = mu*buf + d;  p -= lr*buf   (buf starts at 0, so the
// first step's "buf = d" special case is the same arithmetic).  One launch for ALL parameters.
__global__ __launch_bounds__(256) void k_sgd(long n, float* __restrict__ p, const float* __restrict__ g,
                                             float* __restrict__ buf, const float* __restrict__ lr_ptr, float lr,
                                             float mu, float wd, float gscale) {
  const float rate = lr_ptr ? *lr_ptr : lr;
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 bb = reinterpret_cast<float4*>(buf)[i];
    bb.x = mu * bb.x + (gg.x * gscale + wd * pp.x); pp.x -= rate * bb.x;
    bb.y = mu * bb.y + (gg.y * gscale + wd * pp.y); pp.y -= rate * bb.y;
    bb.z = mu * bb.z + (gg.z * gscale + wd * pp.z); pp.z -= rate * bb.z;
    bb.w = mu * bb.w + (gg.w * gscale + wd * pp.w); pp.w -= rate * bb.w;
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(buf)[i] = bb;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (nv << 2) + threadIdx.x;
    const float b = mu * buf[i] + (g[i] * gscale + wd * p[i]);
    buf[i] = b;
    p[i] -= rate * b;
  }
}

// master fp32 KRSC -> kernel copies: KRSC in T (forward / wgrad layout) and CRSK in T (dgrad)
template <typename T>
__global__ __launch_bounds__(256) void k_weight_prep(int Co, int RS, int Ci, const float* __restrict__ w,
                                                     T* __restrict__ krsc, T* __restrict__ crsk) {
  const long total = (long)Co * RS * Ci;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float v = w[i];
    if (krsc) krsc[i] = (T)v;
    if (crsk) {
      const int ci = (int)(i % Ci);
      const long t = i / Ci;
      const int rs = (int)(t % RS);
      const int co = (int)(t / RS);
      crsk[((long)ci * RS + rs) * Co + co] = (T)v;
    }
  }
}

// All layers in ONE launch: table [n][8] of int64 = {src offset (floats), Co, RS, Ci, krsc ptr, crsk ptr,
// first block, mode}.  mode 0: a block copies 1024 consecutive elements (no transpose: the stem).
// mode 1: a block owns one 64(co) x 64(ci) tile of one tap; the CRSK copy goes through an LDS transpose so
// both copies are written in full 8-byte-per-lane rows (element-wise scattered bf16 stores cost ~10x the
// HBM write traffic: measured 991 MB per launch for 98 MB of weights).
template <typename T>
__global__ __launch_bounds__(256) void k_weight_prep_batched(int n, const int64_t* __restrict__ table,
                                                             const float* __restrict__ master) {
  __shared__ float tile[64][65];
  int e = 0;
  for (int i = 1; i < n; ++i) e = ((int64_t)blockIdx.x >= table[i * 8 + 6]) ? i : e;   // tables are ~55 entries
  const int64_t* t = table + e * 8;
  const float* w = master + t[0];
  const int Co = (int)t[1], RS = (int)t[2], Ci = (int)t[3];
  T* krsc = reinterpret_cast<T*>(t[4]);
  T* crsk = reinterpret_cast<T*>(t[5]);
  const long local = (long)blockIdx.x - t[6];
  if (t[7] == 0) {
    const long total = (long)Co * RS * Ci;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long i = local * 1024 + threadIdx.x + 256 * k;
      if (i < total && krsc) krsc[i] = (T)w[i];
    }
    return;
  }
  const int tci = Ci / 64, tco = Co / 64;
  const int ci_t = (int)(local % tci), co_t = (int)((local / tci) % tco), rs = (int)(local / ((long)tci * tco));
  const int r = threadIdx.x >> 4, q = (threadIdx.x & 15) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = co_t * 64 + r + 16 * k, ci = ci_t * 64 + q;
    const long idx = ((long)co * RS + rs) * Ci + ci;
    const float4 v = *reinterpret_cast<const float4*>(w + idx);
    tile[r + 16 * k][q] = v.x; tile[r + 16 * k][q + 1] = v.y; tile[r + 16 * k][q + 2] = v.z; tile[r + 16 * k][q + 3] = v.w;
    if (krsc) { krsc[idx] = (T)v.x; krsc[idx + 1] = (T)v.y; krsc[idx + 2] = (T)v.z; krsc[idx + 3] = (T)v.w; }
  }
  __syncthreads();
  if (crsk) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ci = ci_t * 64 + r + 16 * k, co = co_t * 64 + q;
      const long idx = ((long)ci * RS + rs) * Co + co;
      crsk[idx] = (T)tile[q][r + 16 * k]; crsk[idx + 1] = (T)tile[q + 1][r + 16 * k];
      crsk[idx + 2] = (T)tile[q + 2][r + 16 * k]; crsk[idx + 3] = (T)tile[q + 3][r + 16 * k];
    }
  }
}

// The optimiser step AND the kernel-format weight copies in ONE launch (round 4; replaces k_sgd + k_weight_prep_batched +
// the gradient zero-fill of the next step: 150 us and 0.5 GB per step).  The table is k_weight_prep_batched's, extended by
// mode 2 rows for the ranges that have no kernel-format copy (BatchNorm gamma / beta, fc bias, the margin head):
//   mode 0: a block updates 1024 consecutive elements and writes them as T (the stem's [64][7][8][4]);
//   mode 1: a block owns one 64(co) x 64(ci) tile of one tap: SGD on the fp32 master, KRSC copy straight from registers,
//           CRSK copy through an LDS transpose;
//   mode 2: a block updates 1024 consecutive elements {src offset, length}.
// Every parameter belongs to exactly one block.  SGD arithmetic is k_sgd's, expression for expression.  `zero_g`: the
// gradient it has just consumed is zeroed (the next step's backward accumulates into it).
template <typename T>
__global__ __launch_bounds__(256) void k_sgd_prep(int n, const int64_t* __restrict__ table, float* __restrict__ p,
                                                  float* __restrict__ g, float* __restrict__ buf,
                                                  const float* __restrict__ lr_ptr, float lr, float mu, float wd, float gscale,
                                                  int zero_g) {
  __shared__ float tile[64][65];
  const float rate = lr_ptr ? *lr_ptr : lr;
  int e = 0;
  for (int i = 1; i < n; ++i) e = ((int64_t)blockIdx.x >= table[i * 8 + 6]) ? i : e;
  const int64_t* t = table + e * 8;
  const long base = t[0];
  const long local = (long)blockIdx.x - t[6];
  auto upd = [&](long idx) -> float4 {
    float4 pp = *reinterpret_cast<float4*>(p + idx);
    const float4 gg = *reinterpret_cast<const float4*>(g + idx);
    float4 bb = *reinterpret_cast<float4*>(buf + idx);
    bb.x = mu * bb.x + (gg.x * gscale + wd * pp.x); pp.x -= rate * bb.x;
    bb.y = mu * bb.y + (gg.y * gscale + wd * pp.y); pp.y -= rate * bb.y;
    bb.z = mu * bb.z + (gg.z * gscale + wd * pp.z); pp.z -= rate * bb.z;
    bb.w = mu * bb.w + (gg.w * gscale + wd * pp.w); pp.w -= rate * bb.w;
    *reinterpret_cast<float4*>(p + idx) = pp;
    *reinterpret_cast<float4*>(buf + idx) = bb;
    if (zero_g) *reinterpret_cast<float4*>(g + idx) = make_float4(0.f, 0.f, 0.f, 0.f);
    return pp;
  };
  auto store4 = [](T* dst, float4 v) {
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(dst) = v;
    else {
      typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
      *reinterpret_cast<bf16x4_t*>(dst) = o;
    }
  };
  if (t[7] != 1) {
    const long total = t[7] == 0 ? t[1] * t[2] * t[3] : t[1];
    T* krsc = t[7] == 0 ? reinterpret_cast<T*>(t[4]) : nullptr;
    const long i = local * 1024 + 4 * threadIdx.x;
    if (i < total) {                     // (ranges are multiples of 4 elements: checked on the host)
      const float4 v = upd(base + i);
      if (krsc) store4(krsc + i, v);
    }
    return;
  }
  const int Co = (int)t[1], RS = (int)t[2], Ci = (int)t[3];
  T* krsc = reinterpret_cast<T*>(t[4]);
  T* crsk = reinterpret_cast<T*>(t[5]);
  const int tci = Ci / 64, tco = Co / 64;
  const int ci_t = (int)(local % tci), co_t = (int)((local / tci) % tco), rs = (int)(local / ((long)tci * tco));
  const int r = threadIdx.x >> 4, q = (threadIdx.x & 15) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = co_t * 64 + r + 16 * k, ci = ci_t * 64 + q;
    const long idx = ((long)co * RS + rs) * Ci + ci;
    const float4 v = upd(base + idx);
    tile[r + 16 * k][q] = v.x; tile[r + 16 * k][q + 1] = v.y; tile[r + 16 * k][q + 2] = v.z; tile[r + 16 * k][q + 3] = v.w;
    if (krsc) store4(krsc + idx, v);
  }
  __syncthreads();
  if (crsk) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ci = ci_t * 64 + r + 16 * k, co = co_t * 64 + q;
      const long idx = ((long)ci * RS + rs) * Co + co;
      store4(crsk + idx, make_float4(tile[q][r + 16 * k], tile[q + 1][r + 16 * k], tile[q + 2][r + 16 * k], tile[q + 3][r + 16 * k]));
    }
  }
}

// fp32 NCHW image batch (already normalised to [-1,1]) -> zero-bordered NHWC4 in T for the stem
template <typename T>
__global__ __launch_bounds__(256) void k_input_prep_f32(int N, int H, int W, int Hp, int Wp,
                                                        const float* __restrict__ x, T* __restrict__ out) {
  const long total = (long)N * Hp * Wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int wp = (int)(i % Wp);
    long t = i / Wp;
    const int hp = (int)(t % Hp);
    const int n = (int)(t / Hp);
    const int h = hp - 3, w = wp - 3;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = x[(((long)n * 3 + c) * H + h) * W + w];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) out[i * 4 + c] = (T)v[c];
  }
}

// uint8 NHWC (decoder output) -> same layout, fused ToTensor + Normalize(0.5,0.5): x/127.5 - 1
template <typename T>
__global__ __launch_bounds__(256) void k_input_prep_u8(int N, int H, int W, int Hp, int Wp,
                                                       const uint8_t* __restrict__ x, T* __restrict__ out) {
  const long total = (long)N * Hp * Wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int wp = (int)(i % Wp);
    long t = i / Wp;
    const int hp = (int)(t % Hp);
    const int n = (int)(t / Hp);
    const int h = hp - 3, w = wp - 3;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
      const uint8_t* px = x + (((long)n * H + h) * W + w) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = ((float)px[c] / 255.f - 0.5f) / 0.5f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) out[i * 4 + c] = (T)v[c];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_cast_from_f32(long n, const float* __restrict__ x, T* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (T)x[i];
}
template <typename T>
__global__ __launch_bounds__(256) void k_cast_to_f32(long n, const T* __restrict__ x, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (float)x[i];
}

// out[c] += sum_r x[r][c]   (fc bias gradient).  16 columns x 16 row-lanes per block, 8 independent loads per
// lane in flight (a single thread walking 256 rows is 256 dependent memory round trips: 52 us for 0.5 MB).
__global__ __launch_bounds__(256) void k_colsum_f32(int rows, int C, const float* __restrict__ x, float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C) {
    for (int r0 = rl; r0 < rows; r0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int r = r0 + 16 * u; v[u] = r < rows ? x[(long)r * C + c] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    out[c] += t;
  }
}

__global__ __launch_bounds__(256) void k_copy_f32m(long n, const float* __restrict__ x, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = x[i];
}

static inline int ew_grid2(long n) {
  long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace frx
using namespace frx;

extern "C" int frx_sgd_step(int device, frx_stream_t stream, int64_t n, float* p, const float* g, float* buf,
                            const float* lr_dev, float lr, float momentum, float weight_decay, float grad_scale) {
  FRX_CHECK_ARG(n >= 0, "sgd_step: n<0");
  if (n == 0) return FRX_OK;
  FRX_CHECK_ARG(p && g && buf, "sgd_step: NULL pointer");
  FRX_CHECK_ARG((((size_t)p | (size_t)g | (size_t)buf) & 15) == 0, "sgd_step: buffers must be 16-byte aligned");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_sgd, dim3(ew_grid2(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, (long)n, p, g, buf, lr_dev,
                     lr, momentum, weight_decay, grad_scale);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_weight_prep(int device, frx_stream_t stream, int dtype, int Co, int RS, int Ci, const float* master,
                               void* krsc, void* crsk) {
  FRX_CHECK_ARG(dtype == FRX_F32 || dtype == FRX_BF16, "weight_prep: dtype");
  FRX_CHECK_ARG(master && (krsc || crsk) && Co > 0 && RS > 0 && Ci > 0, "weight_prep: bad args");
  FRX_ENTER(device);
  const long total = (long)Co * RS * Ci;
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_weight_prep<bf16_t>, dim3(ew_grid2(total)), dim3(256), 0, (hipStream_t)stream, Co, RS, Ci,
                       master, (bf16_t*)krsc, (bf16_t*)crsk);
  else
    hipLaunchKernelGGL(k_weight_prep<float>, dim3(ew_grid2(total)), dim3(256), 0, (hipStream_t)stream, Co, RS, Ci,
                       master, (float*)krsc, (float*)crsk);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_input_prep(int device, frx_stream_t stream, int dtype, int N, int H, int W, const void* images,
                              int is_u8_nhwc, void* out, int64_t out_elems) {
  FRX_CHECK_ARG(dtype == FRX_F32 || dtype == FRX_BF16, "input_prep: dtype");
  FRX_CHECK_ARG(images && out && N > 0 && H > 0 && W > 0, "input_prep: bad args");
  int hp, wp;
  frx_stem_padded_dims(H, W, &hp, &wp);
  const long total = (long)N * hp * wp;
  FRX_CHECK_ARG(out_elems == (int64_t)total * 4, "input_prep: a %dx%dx%d batch needs %ld output elements, the destination holds %ld",
                N, H, W, total * 4, (long)out_elems);
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  if (is_u8_nhwc) {
    if (dtype == FRX_BF16) hipLaunchKernelGGL(k_input_prep_u8<bf16_t>, dim3(ew_grid2(total)), dim3(256), 0, st, N, H, W, hp, wp, (const uint8_t*)images, (bf16_t*)out);
    else hipLaunchKernelGGL(k_input_prep_u8<float>, dim3(ew_grid2(total)), dim3(256), 0, st, N, H, W, hp, wp, (const uint8_t*)images, (float*)out);
  } else {
    if (dtype == FRX_BF16) hipLaunchKernelGGL(k_input_prep_f32<bf16_t>, dim3(ew_grid2(total)), dim3(256), 0, st, N, H, W, hp, wp, (const float*)images, (bf16_t*)out);
    else hipLaunchKernelGGL(k_input_prep_f32<float>, dim3(ew_grid2(total)), dim3(256), 0, st, N, H, W, hp, wp, (const float*)images, (float*)out);
  }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_cast(int device, frx_stream_t stream, int dtype, int to_f32, int64_t n, const void* x, void* y) {
  FRX_CHECK_ARG(dtype == FRX_F32 || dtype == FRX_BF16, "cast: dtype");
  FRX_CHECK_ARG(n >= 0, "cast: n<0");
  if (n == 0) return FRX_OK;
  FRX_CHECK_ARG(x && y, "cast: NULL pointer");
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRX_F32) {
    hipLaunchKernelGGL(k_copy_f32m, dim3(ew_grid2(n)), dim3(256), 0, st, (long)n, (const float*)x, (float*)y);   // (no memcpy node in a captured step)
    FRX_LAUNCH_CHECK();
    return FRX_OK;
  }
  if (to_f32) hipLaunchKernelGGL(k_cast_to_f32<bf16_t>, dim3(ew_grid2(n)), dim3(256), 0, st, (long)n, (const bf16_t*)x, (float*)y);
  else hipLaunchKernelGGL(k_cast_from_f32<bf16_t>, dim3(ew_grid2(n)), dim3(256), 0, st, (long)n, (const float*)x, (bf16_t*)y);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_colsum_f32(int device, frx_stream_t stream, int rows, int C, const float* x, float* out) {
  FRX_CHECK_ARG(x && out && rows > 0 && C > 0, "colsum: bad args");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_colsum_f32, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, rows, C, x, out);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_weight_prep_batched(int device, frx_stream_t stream, int dtype, int n, const int64_t* table_dev,
                                       const float* master, int total_blocks) {
  FRX_CHECK_ARG(dtype == FRX_F32 || dtype == FRX_BF16, "weight_prep_batched: dtype");
  FRX_CHECK_ARG(n > 0 && table_dev && master && total_blocks > 0, "weight_prep_batched: bad args");
  FRX_ENTER(device);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_weight_prep_batched<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, n, table_dev, master);
  else
    hipLaunchKernelGGL(k_weight_prep_batched<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, n, table_dev, master);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_sgd_step_prep(int device, frx_stream_t stream, int dtype, int n, const int64_t* table_dev, int total_blocks,
                                 float* p, float* g, float* buf, const float* lr_dev, float lr, float momentum,
                                 float weight_decay, float grad_scale, int zero_grads) {
  FRX_CHECK_ARG(dtype == FRX_F32 || dtype == FRX_BF16, "sgd_step_prep: dtype");
  FRX_CHECK_ARG(n > 0 && table_dev && total_blocks > 0 && p && g && buf, "sgd_step_prep: bad args");
  FRX_CHECK_ARG((((size_t)p | (size_t)g | (size_t)buf) & 15) == 0, "sgd_step_prep: buffers must be 16-byte aligned");
  FRX_ENTER(device);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_sgd_prep<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, n, table_dev, p, g, buf, lr_dev, lr,
                       momentum, weight_decay, grad_scale, zero_grads);
  else
    hipLaunchKernelGGL(k_sgd_prep<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, n, table_dev, p, g, buf, lr_dev, lr,
                       momentum, weight_decay, grad_scale, zero_grads);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
