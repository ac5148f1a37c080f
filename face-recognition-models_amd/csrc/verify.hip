// LFW-style pair verification: pairwise cosine of L2-normalised embeddings and the threshold
// count.  Reference: main_code/utils/model_utils.py:370-375 (evaluate), :391-393
// (tune_threshold_roc).  HBM-bound: 2*P*D*4 bytes read, 4*P written.
#include "frx_common.h"

namespace frx {

// One wavefront per pair: three reductions (a.b, |a|^2, |b|^2) in one pass over 2 x D floats.
__global__ __launch_bounds__(256) void k_pair_cosine(const float* __restrict__ f1,
                                                     const float* __restrict__ f2, long P, int D,
                                                     float* __restrict__ out) {
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= P) return;
  const int lane = threadIdx.x & 63;
  const float* a = f1 + pair * D;
  const float* b = f2 + pair * D;
  float ab = 0.f, aa = 0.f, bb = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 x = *reinterpret_cast<const float4*>(a + d);
    const float4 y = *reinterpret_cast<const float4*>(b + d);
    ab += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    aa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    bb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
  }
  ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
  if (lane == 0) out[pair] = ab / (fmaxf(sqrtf(aa), 1e-12f) * fmaxf(sqrtf(bb), 1e-12f));
}

__global__ __launch_bounds__(256) void k_threshold_count(const float* __restrict__ cos,
                                                         const int64_t* __restrict__ same, long P,
                                                         float thr, int32_t* __restrict__ correct) {
  int c = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long)gridDim.x * 256)
    c += ((cos[i] > thr) ? 1 : 0) == (int)same[i];
  c = wave_sum_i(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(correct, c);
}

}  // namespace frx
using namespace frx;

extern "C" int frx_pair_cosine(int device, frx_stream_t stream, const float* f1, const float* f2,
                               int64_t P, int32_t D, float* cos_out) {
  FRX_CHECK_ARG(P >= 0 && D > 0 && D % 4 == 0, "pair_cosine: need P>=0 and D%%4==0 (P=%lld D=%d)", (long long)P, D);
  if (P == 0) return FRX_OK;
  FRX_CHECK_ARG(f1 && f2 && cos_out, "pair_cosine: NULL pointer");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_pair_cosine, dim3(cdiv(P, 4)), dim3(256), 0, (hipStream_t)stream, f1, f2, (long)P, D, cos_out);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_threshold_count(int device, frx_stream_t stream, const float* cos, const int64_t* same,
                                   int64_t P, float thr, int32_t* correct) {
  FRX_CHECK_ARG(P >= 0, "threshold_count: P<0");
  if (P == 0) return FRX_OK;
  FRX_CHECK_ARG(cos && same && correct, "threshold_count: NULL pointer");
  FRX_ENTER(device);
  const int blocks = (int)(cdiv(P, 256) > 1024 ? 1024 : cdiv(P, 256));
  hipLaunchKernelGGL(k_threshold_count, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cos, same, (long)P, thr, correct);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
