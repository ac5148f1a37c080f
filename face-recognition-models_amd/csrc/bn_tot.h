// Train-mode BatchNorm statistics as REPLICATED TOTAL ROWS (round 3).
//
// The per-tile partial rows [tilesM][2][C] + one finalize launch per BatchNorm layer cost a training step 106 dependent
// launches (0.56 ms of 6.9, DESIGN 8.2: a launch behind a big kernel costs ~5 us, and two in-kernel hand-off designs
// were slower still).  Here the PRODUCING kernel adds its per-channel sums with float atomics into R replicated rows
// tot[R][2][C] (replica = row-tile index mod R: a few dozen adders per address, the regime the atomics run at full rate
// in), the launch boundary that exists anyway is the only synchronisation, and every CONSUMER block derives the
// per-channel constants it needs from the R rows while it sets up.  One batched launch per pass turns the totals into the
// canonical per-channel arrays (mean / invstd / scale / shift / running statistics; dgamma / dbeta / coefficients) for the
// kernels that run much later, and zeroes the rows for the next step.
//
// Float atomics make the sums depend on arrival order in the last bits: this path serves the bf16 speed mode; the fp32
// parity mode and FRX_BN_DETERMINISTIC=1 keep the partial rows + finalize launches (bit-reproducible).
//
// Every consumer runs the SAME reduction (row order 0..R-1 in double) and the same closing arithmetic as
// k_bn_finalize / k_bn_bwd_finalize, so the constants it derives are the ones the batched launch writes.
#pragma once
#include "frx_common.h"

namespace frx {

struct BnTot {              // device-side view of frx_bn_tot
  const float* tot;         // [R][2][C]
  const float* gamma;       // [C]
  const float* beta;        // [C]   forward
  const float* mean;        // [C]   backward
  const float* invstd;      // [C]   backward
  int R;
  float eps;
  double inv_count;         // 1 / elements per channel
};

static inline BnTot bn_tot_arg(const frx_bn_tot* t) {
  BnTot b{};
  if (t) { b.tot = t->totals; b.gamma = t->gamma; b.beta = t->beta; b.mean = t->mean; b.invstd = t->invstd; b.R = t->replicas; b.inv_count = 1.0 / (double)t->count; b.eps = t->eps; }
  return b;
}

// field-wise copy (the source may live in another address space: the kernarg segment)
template <typename S> __device__ __forceinline__ BnTot bn_tot_copy(const S& s) {
  BnTot b;
  b.tot = s.tot; b.gamma = s.gamma; b.beta = s.beta; b.mean = s.mean; b.invstd = s.invstd; b.R = s.R; b.eps = s.eps; b.inv_count = s.inv_count;
  return b;
}

// The closing arithmetic of BatchNorm statistics, shared by EVERY kernel that turns sums into constants (the per-layer
// finalize kernels, the batched ones, and each consumer's prologue): one definition, so they agree bit for bit.
// forward: channel sums (s, q) of y and y^2 -> mean, biased variance (double), invstd = rsqrt(var + eps) in float (one
// Newton step on v_rsq_f32: < 1 ulp, and a fraction of the double sqrt + divide it replaces in every consumer block),
// scale = gamma * invstd, shift = beta - mean * scale
__device__ __forceinline__ void bn_fwd_consts(double s, double q, double inv_count, float gamma, float beta, float eps,
                                              float& mean_o, float& invstd_o, float& scale_o, float& shift_o, double& var_o) {
  const double mean = s * inv_count;
  double var = q * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float vf = (float)var + eps;
  float r = __frsqrt_rn(vf);
  r = r * (1.5f - 0.5f * vf * r * r);
  const float sc = gamma * r;
  mean_o = (float)mean; invstd_o = r; scale_o = sc; shift_o = beta - (float)mean * sc; var_o = var;
}
// backward: (a, b) = (sum dz, sum dz*xhat) -> dy = alpha*dz + beta*y + gam with alpha = gamma*invstd,
// beta = -alpha*invstd*mean(dz*xhat), gam = alpha*(mu*invstd*mean(dz*xhat) - mean(dz))
__device__ __forceinline__ void bn_bwd_consts(double a, double b, double inv_count, float gamma, float mean, float invstd,
                                              float& al, float& be, float& ga) {
  const double k1 = (double)gamma * (double)invstd;
  const double c1 = a * inv_count, c2 = b * inv_count;
  const double is = (double)invstd, mu = (double)mean;
  al = (float)k1;
  be = (float)(-k1 * is * c2);
  ga = (float)(k1 * (mu * is * c2 - c1));
}

// The two totals of channel c: rows 0..R-1 summed in double, in row order.  R is a power of two; from 8 rows up the
// sixteen loads of eight rows are in flight together (one memory round trip per eight rows).
__device__ __forceinline__ void bn_tot_sum(const float* __restrict__ tot, int R, int C, int c, double& s, double& q) {
  s = 0.0; q = 0.0;
  if (R >= 8) {
    for (int r0 = 0; r0 < R; r0 += 8) {
      float vs[8], vq[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) { vs[r] = tot[(2 * (r0 + r)) * C + c]; vq[r] = tot[(2 * (r0 + r) + 1) * C + c]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) { s += (double)vs[r]; q += (double)vq[r]; }
    }
  } else {
    float vs[4], vq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {           // (rows past R re-read row 0 and are not added: the loads stay unconditional)
      const int rr = r < R ? r : 0;
      vs[r] = tot[(2 * rr) * C + c]; vq[r] = tot[(2 * rr + 1) * C + c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s += r < R ? (double)vs[r] : 0.0; q += r < R ? (double)vq[r] : 0.0; }
  }
}

// Block-cooperative: every thread of an NT-thread block calls this; emit(c, s, q) runs once per channel c < C with the
// channel's totals (one thread per channel sums all rows in order: every consumer derives bit-identical constants).
template <int NT, typename Emit>
__device__ __forceinline__ void bn_tot_foreach(const float* __restrict__ tot, int R, int C, Emit&& emit) {
  for (int c = threadIdx.x; c < C; c += NT) {
    double s, q;
    bn_tot_sum(tot, R, C, c, s, q);
    emit(c, s, q);
  }
}

}  // namespace frx
