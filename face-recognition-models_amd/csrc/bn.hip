// Train-mode BatchNorm pieces that cannot live inside a conv kernel, the residual merge,
// the stem max-pool and the average pool -- all HBM-bound, 16-byte vectorised, NHWC.
// Reference call sites: nn.BatchNorm2d x53, nn.ReLU x49, residual add x16, MaxPool2d,
// AdaptiveAvgPool2d of the torchvision ResNet-50 built in main_code/utils/backbones.py:16-18.
#include "conv_kernels.h"
#include <stdlib.h>

namespace frx {

template <typename T> struct Vec16 {
  uint4 raw;
  static constexpr int N = 16 / sizeof(T);
  __device__ __forceinline__ float get(int j) const {
    if constexpr (sizeof(T) == 4) return reinterpret_cast<const float*>(&raw)[j];
    else return (float)reinterpret_cast<const bf16_t*>(&raw)[j];
  }
  __device__ __forceinline__ void set(int j, float v) {
    if constexpr (sizeof(T) == 4) reinterpret_cast<float*>(&raw)[j] = v;
    else reinterpret_cast<bf16_t*>(&raw)[j] = (bf16_t)v;
  }
};

// ---------------------------------------------------------------- BN forward statistics
// partial [rows][2][C] (sum, sum of squares from the conv epilogue) -> batch mean / biased var,
// scale = gamma*invstd, shift = beta - mean*scale, running stats (momentum 0.1, unbiased var).
// Column sums of a [rows][2][C] partial buffer.  The buffer was just written by up to thousands of
// blocks, so this is latency-bound: 16 channels x 64 row-lanes per block, each lane with 8 rows (16 loads)
// in flight; fp32 partial runs are short, the total is fp64.
constexpr int FIN_THREADS = 1024, FIN_RL = FIN_THREADS / 16;
__device__ __forceinline__ void partial_colsum(const float* __restrict__ partial, int rows, int C,
                                               double& s, double& q, int& c) {
  __shared__ double red[2][FIN_RL][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  c = blockIdx.x * 16 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    constexpr int U = 8;
    for (int r0 = rl; r0 < rows; r0 += U * FIN_RL) {
      float va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + u * FIN_RL;
        const long o = (long)(r < rows ? r : r0) * 2 * C + c;
        va[u] = partial[o]; vb[u] = partial[o + C];
      }
      float fa = 0.f, fb = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = r0 + u * FIN_RL < rows;
        fa += live ? va[u] : 0.f; fb += live ? vb[u] : 0.f;
      }
      a += (double)fa; b += (double)fb;
    }
  }
  red[0][rl][cl] = a; red[1][rl][cl] = b;
  __syncthreads();
  // 64 row-lanes -> 4 (the first wave: 16 channels x 4 groups of 16 row-lanes) -> 1
  __shared__ double red2[2][4][16];
  if (rl < 4) {
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ss += red[0][rl * 16 + i][cl]; qq += red[1][rl * 16 + i][cl]; }
    red2[0][rl][cl] = ss; red2[1][rl][cl] = qq;
  }
  __syncthreads();
  s = 0.0; q = 0.0;
  if (threadIdx.x < 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += red2[0][i][cl]; q += red2[1][i][cl]; }
  }
}

__global__ __launch_bounds__(FIN_THREADS) void k_bn_finalize(const float* __restrict__ partial, int rows, int C,
                                                     double count, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float momentum,
                                                     float* __restrict__ rmean, float* __restrict__ rvar,
                                                     float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                     float* __restrict__ scale_o, float* __restrict__ shift_o) {
  double s, q;
  int c;
  partial_colsum(partial, rows, C, s, q, c);
  if (threadIdx.x >= 16 || c >= C) return;
  float mean, invstd, sc, sh; double var;
  bn_fwd_consts(s, q, 1.0 / count, gamma[c], beta[c], eps, mean, invstd, sc, sh, var);
  mean_o[c] = mean;
  invstd_o[c] = invstd;
  scale_o[c] = sc;
  shift_o[c] = sh;
  if (rmean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

__global__ __launch_bounds__(256) void k_bn_eval_affine(int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ rmean,
                                                        const float* __restrict__ rvar, float eps,
                                                        float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rvar[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rmean[c] * sc;
}

// ---------------------------------------------------------------- residual merge (forward)
// out = relu(s3*y3 + b3 + idn)            idn = block input, or sd*yd + bd for a downsample branch
// Row-major [rows][C] sweeps: a thread keeps ONE 16-byte channel group and walks down the rows, so
// the per-channel constants sit in registers instead of being re-fetched per element.
struct RowWalk {
  int groups, gpb, rpp, tg, trow;
  __device__ __forceinline__ RowWalk(int C, int V) {
    groups = C / V;
    gpb = groups < 256 ? groups : 256;
    rpp = 256 / gpb;
    tg = threadIdx.x % gpb;
    trow = threadIdx.x / gpb;
  }
};

// 8 consecutive per-channel constants as two 16-byte loads (V = 4: one)
template <int V> __device__ __forceinline__ void load_consts(const float* __restrict__ p, float (&o)[V]) {
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
    o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
  }
}
template <int V> __device__ __forceinline__ void fill_consts(float v, float (&o)[V]) {
#pragma unroll
  for (int j = 0; j < V; ++j) o[j] = v;
}

// The row loops below are unrolled RU-fold with every load of the RU rows issued before the first use:
// one 16-byte load pair per thread in flight leaves these sweeps latency-bound at ~2.6 TB/s.
constexpr int RU = 4;

// Per-channel constants derived from replicated totals (bn_tot.h) are staged in dynamic LDS, [table][C] floats
extern __shared__ __attribute__((aligned(16))) float s_bn_dyn[];
__device__ __forceinline__ void stage_fwd_consts(const BnTot& b, int C, float* lsc, float* lsh) {
  bn_tot_foreach<256>(b.tot, b.R, C, [&](int c, double sm, double sq) {
    float mean, invstd, sc, sh; double var;
    bn_fwd_consts(sm, sq, b.inv_count, b.gamma[c], b.beta[c], b.eps, mean, invstd, sc, sh, var);
    lsc[c] = sc; lsh[c] = sh;
  });
}
__device__ __forceinline__ void stage_bwd_consts(const BnTot& b, int C, float* lcoef /* [3][C] */) {
  bn_tot_foreach<256>(b.tot, b.R, C, [&](int c, double sa, double sb) {
    float al, be, ga;
    bn_bwd_consts(sa, sb, b.inv_count, b.gamma[c], b.mean[c], b.invstd[c], al, be, ga);
    lcoef[c] = al; lcoef[C + c] = be; lcoef[2 * C + c] = ga;
  });
}

template <typename T>
__global__ __launch_bounds__(256) void k_merge_fwd(long rows, int C, const T* __restrict__ y3,
                                                   const float* s3, const float* b3,
                                                   const T* __restrict__ idn, const float* sd,
                                                   const float* bd, T* __restrict__ out,
                                                   uint8_t* __restrict__ mask, BnTot t3, BnTot td) {
  constexpr int V = Vec16<T>::N;
  const RowWalk w(C, V);
  const long stride = (long)gridDim.x * w.rpp;
  const long r_first = (long)blockIdx.x * w.rpp + w.trow;
  // the first rows' loads are issued BEFORE the constants are derived from the totals: the derivation (R rows per channel,
  // closing arithmetic, an LDS round trip) then runs under their latency instead of in front of every block
  Vec16<T> a0[RU], b0[RU];
  if (t3.tot) {               // (block-uniform) constants from the producers' totals instead of finalized arrays
    if (r_first < rows) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const long rr = r_first + u * stride;
        const long i = (rr < rows ? rr : r_first) * w.groups + w.tg;
        a0[u].raw = reinterpret_cast<const uint4*>(y3)[i];
        b0[u].raw = reinterpret_cast<const uint4*>(idn)[i];
      }
    }
    stage_fwd_consts(t3, C, s_bn_dyn, s_bn_dyn + C);
    s3 = s_bn_dyn; b3 = s_bn_dyn + C;
    if (td.tot) {
      stage_fwd_consts(td, C, s_bn_dyn + 2 * C, s_bn_dyn + 3 * C);
      sd = s_bn_dyn + 2 * C; bd = s_bn_dyn + 3 * C;
    }
    __syncthreads();
  }
  for (int g0 = 0; g0 < w.groups; g0 += w.gpb) {
    const int grp = g0 + w.tg, c = grp * V;
    float ks[V], kb[V], ds[V], db[V];
    load_consts<V>(s3 + c, ks); load_consts<V>(b3 + c, kb);
    if (sd) { load_consts<V>(sd + c, ds); load_consts<V>(bd + c, db); } else { fill_consts<V>(1.f, ds); fill_consts<V>(0.f, db); }
    for (long r = r_first; r < rows; r += RU * stride) {
      Vec16<T> a[RU], b[RU];
      if (t3.tot && g0 == 0 && r == r_first) {
#pragma unroll
        for (int u = 0; u < RU; ++u) { a[u] = a0[u]; b[u] = b0[u]; }
      } else {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const long rr = r + u * stride;
          const long i = (rr < rows ? rr : r) * w.groups + grp;      // clamp: loads stay unconditional
          a[u].raw = reinterpret_cast<const uint4*>(y3)[i];
          b[u].raw = reinterpret_cast<const uint4*>(idn)[i];
        }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const long rr = r + u * stride;
        Vec16<T> o;
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          // (fma for fma the expression of conv_kernels.h: merge_vec -- a consumer that evaluates the merge in its prologue
          // (frx_conv_fwd_merge) stages exactly the value this pass writes)
          o.set(j, fmaxf(fmaf(a[u].get(j), ks[j], fmaf(b[u].get(j), ds[j], kb[j] + db[j])), 0.f));
          bits |= (o.get(j) > 0.f ? 1u : 0u) << j;      // of the ROUNDED output: what `out > 0` would see
        }
        if (rr < rows) {
          reinterpret_cast<uint4*>(out)[rr * w.groups + grp] = o.raw;
          if (mask) mask[rr * w.groups + grp] = (uint8_t)bits;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- BN backward
// dz = g * mask;  mask = (out > 0) when `out` is given (merge ReLU), else (scale*y+shift > 0) when
// relu, else 1.  Accumulates per channel  sum(dz)  and  sum(dz * xhat),  xhat = (y-mean)*invstd.
// Block b handles rows b, b+gridDim, ...; thread owns one 16-byte channel group.
template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(long rows, int C, const T* __restrict__ g,
                                                       const T* __restrict__ y, const T* __restrict__ out,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, T* __restrict__ dz_out,
                                                       float* __restrict__ partial, int g_pool_hw, int tot_R) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][256][V + 1];
  // g_pool_hw > 0: `g` is the gradient of an average pool over g_pool_hw pixels, [rows / g_pool_hw][C]; each row takes
  // its image's entry / g_pool_hw, rounded to T as the stand-alone frx_avgpool_bwd stored it
  const float ginv = g_pool_hw > 0 ? 1.f / (float)g_pool_hw : 1.f;
  const int groups = C / V;                         // channel groups per row
  const int gpb = groups < 256 ? groups : 256;      // groups handled per pass
  const int rpp = 256 / gpb;                        // rows per pass
  const int tg = threadIdx.x % gpb, trow = threadIdx.x / gpb;
  for (int g0 = 0; g0 < groups; g0 += gpb) {
    const int grp = g0 + tg;
    const int c = grp * V;
    float s1[V], s2[V], mu[V], is[V], sc[V], sh[V];
    fill_consts<V>(0.f, s1); fill_consts<V>(0.f, s2);
    load_consts<V>(mean + c, mu); load_consts<V>(invstd + c, is);
    if (scale) load_consts<V>(scale + c, sc); else fill_consts<V>(1.f, sc);
    if (shift) load_consts<V>(shift + c, sh); else fill_consts<V>(0.f, sh);
    if (trow < rpp) {
      const long stride = (long)gridDim.x * rpp;
      for (long r = (long)blockIdx.x * rpp + trow; r < rows; r += RU * stride) {
        Vec16<T> vg[RU], vy[RU], vo[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const long rr = r + u * stride;
          const long rc = rr < rows ? rr : r, i = rc * groups + grp;
          vg[u].raw = reinterpret_cast<const uint4*>(g)[g_pool_hw > 0 ? (rc / g_pool_hw) * groups + grp : i];
          vy[u].raw = reinterpret_cast<const uint4*>(y)[i];
          if (out) vo[u].raw = reinterpret_cast<const uint4*>(out)[i];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const long rr = r + u * stride;
          const bool live = rr < rows;
          if (g_pool_hw > 0) {
#pragma unroll
            for (int j = 0; j < V; ++j) vg[u].set(j, vg[u].get(j) * ginv);
          }
          Vec16<T> vz;
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float yy = vy[u].get(j);
            bool on = live;
            if (out) on = on && vo[u].get(j) > 0.f;
            else if (relu) on = on && fmaf(yy, sc[j], sh[j]) > 0.f;
            const float dz = on ? vg[u].get(j) : 0.f;
            s1[j] += dz;
            s2[j] += dz * (yy - mu[j]) * is[j];
            vz.set(j, dz);
          }
          if (dz_out && live) reinterpret_cast<uint4*>(dz_out)[rr * groups + grp] = vz.raw;
        }
      }
    }
    // reduce over the rpp row-lanes that share a channel group
    __syncthreads();
#pragma unroll
    for (int j = 0; j < V; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
    __syncthreads();
    float fa[V], fb[V];
    if (threadIdx.x < gpb) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float a = 0.f, b = 0.f;
        for (int rr = 0; rr < rpp; ++rr) { a += red[0][rr * gpb + threadIdx.x][j]; b += red[1][rr * gpb + threadIdx.x][j]; }
        if (tot_R > 0) { fa[j] = a; fb[j] = b; }      // (added below, one contiguous 256-byte run per wave-instruction)
        else {
          partial[((long)blockIdx.x * 2 + 0) * C + c + j] = a;
          partial[((long)blockIdx.x * 2 + 1) * C + c + j] = b;
        }
      }
    }
    if (tot_R > 0) {        // replicated totals (bn_tot.h): `partial` is [tot_R][2][C], ADDED to.  The sums go through LDS so
      __syncthreads();      // that consecutive lanes add to consecutive channels (the shape float atomics run at full rate in)
      float* flat = &red[0][0][0];                   // 2 * 256 * (V + 1) floats >= 2 * gpb * V
      if (threadIdx.x < gpb) {
#pragma unroll
        for (int j = 0; j < V; ++j) { flat[threadIdx.x * V + j] = fa[j]; flat[gpb * V + threadIdx.x * V + j] = fb[j]; }
      }
      __syncthreads();
      const int span = gpb * V;
      for (int i = threadIdx.x; i < 2 * span; i += 256) {
        const int which = i >= span ? 1 : 0, cc = i - which * span;
        __hip_atomic_fetch_add(partial + ((long)(blockIdx.x % tot_R) * 2 + which) * C + g0 * V + cc, flat[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// partial [nblk][2][C] -> dgamma += , dbeta += , coef [3][C] = (alpha, beta, gam) with dy = alpha*dz + beta*y + gam
__global__ __launch_bounds__(FIN_THREADS) void k_bn_bwd_finalize(const float* __restrict__ partial, int nblk, int C,
                                                         double count, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ coef) {
  double a, b;
  int c;
  partial_colsum(partial, nblk, C, a, b, c);
  if (threadIdx.x >= 16 || c >= C) return;
  if (dbeta) dbeta[c] += (float)a;
  if (dgamma) dgamma[c] += (float)b;
  // dy = k1*(dz - c1 - xhat*c2), xhat = (y-mu)*is   ==   alpha*dz + beta*y + gam   (affine in dz and y)
  float al, be, ga;
  bn_bwd_consts(a, b, 1.0 / count, gamma[c], mean[c], invstd[c], al, be, ga);
  coef[c] = al;
  coef[C + c] = be;
  coef[2 * C + c] = ga;
}

// dy = k1 * (dz - c1 - xhat*c2);  dz = g*mask with the same mask rule as the reduce kernel
template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(long rows, int C, const T* __restrict__ g,
                                                      const T* __restrict__ y, const T* __restrict__ out,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ invstd,
                                                      const float* coef, T* __restrict__ dy, BnTot bt) {
  constexpr int V = Vec16<T>::N;
  const RowWalk w(C, V);
  const long stride = (long)gridDim.x * w.rpp;
  const long r_first = (long)blockIdx.x * w.rpp + w.trow;
  Vec16<T> g0v[RU], y0v[RU];          // (as in k_merge_fwd: the first rows' loads go out before the coefficients are derived)
  if (bt.tot) {
    if (r_first < rows) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const long rr = r_first + u * stride;
        const long i = (rr < rows ? rr : r_first) * w.groups + w.tg;
        g0v[u].raw = reinterpret_cast<const uint4*>(g)[i];
        y0v[u].raw = reinterpret_cast<const uint4*>(y)[i];
      }
    }
    stage_bwd_consts(bt, C, s_bn_dyn);
    coef = s_bn_dyn;
    __syncthreads();
  }
  for (int g0 = 0; g0 < w.groups; g0 += w.gpb) {
    const int grp = g0 + w.tg, c = grp * V;
    float al[V], be[V], ga[V], sc[V], sh[V];
    load_consts<V>(coef + c, al); load_consts<V>(coef + C + c, be); load_consts<V>(coef + 2 * C + c, ga);
    if (relu && !out) { load_consts<V>(scale + c, sc); load_consts<V>(shift + c, sh); } else { fill_consts<V>(1.f, sc); fill_consts<V>(0.f, sh); }
    for (long r = r_first; r < rows; r += RU * stride) {
      Vec16<T> vg[RU], vy[RU], vo[RU];
      const bool pre = bt.tot && g0 == 0 && r == r_first;
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const long rr = r + u * stride;
        const long i = (rr < rows ? rr : r) * w.groups + grp;
        if (pre) { vg[u] = g0v[u]; vy[u] = y0v[u]; }
        else {
          vg[u].raw = reinterpret_cast<const uint4*>(g)[i];
          vy[u].raw = reinterpret_cast<const uint4*>(y)[i];
        }
        if (out) vo[u].raw = reinterpret_cast<const uint4*>(out)[i];
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const long rr = r + u * stride;
        Vec16<T> vd;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float yy = vy[u].get(j);
          bool on = true;
          if (out) on = vo[u].get(j) > 0.f;
          else if (relu) on = fmaf(yy, sc[j], sh[j]) > 0.f;
          const float dz = on ? vg[u].get(j) : 0.f;
          vd.set(j, fmaf(al[j], dz, fmaf(be[j], yy, ga[j])));
        }
        if (rr < rows) reinterpret_cast<uint4*>(dy)[rr * w.groups + grp] = vd.raw;
      }
    }
  }
}

// ---------------------------------------------------------------- stem max-pool 3x3 s2 p1
// out = maxpool(relu(scale*y + shift)); argmax keeps the window position (kh*3+kw) of the first
// maximum in scan order (torch's tie rule) for the backward gather.
template <typename T>
__global__ __launch_bounds__(256) void k_pool_fwd(int N, int H, int W, int C, const T* __restrict__ y,
                                                  const float* scale, const float* shift,
                                                  T* __restrict__ out, uint8_t* __restrict__ argmax, BnTot bt) {
  constexpr int V = Vec16<T>::N;
  if (bt.tot) {
    stage_fwd_consts(bt, C, s_bn_dyn, s_bn_dyn + C);
    scale = s_bn_dyn; shift = s_bn_dyn + C;
    __syncthreads();
  }
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C / V;
  const long total = (long)N * Ho * Wo * G;
  // XCD-contiguous index ranges (block b runs on XCD b % 8): the overlapping 3x3 windows of neighbouring rows are
  // then fetched into ONE L2 instead of up to four (225 -> ~150 MB of HBM traffic for the backward)
  const long chunk = (total + 7) / 8, lo = (long)(blockIdx.x & 7) * chunk, hi = lo + chunk < total ? lo + chunk : total;
  for (long i = lo + (long)(blockIdx.x >> 3) * 256 + threadIdx.x; i < hi; i += (long)(gridDim.x >> 3) * 256) {
    const int g = (int)(i % G);
    long p = i / G;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int c = g * V;
    float sc[V], sh[V], best[V];
    int arg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = scale[c + j]; sh[j] = shift[c + j]; best[j] = -INFINITY; arg[j] = 0; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int h = oh * 2 - 1 + kh, w = ow * 2 - 1 + kw;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
          Vec16<T> v, q;
          v.raw = *reinterpret_cast<const uint4*>(y + (((long)n * H + h) * W + w) * C + c);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            q.set(j, fmaxf(fmaf(v.get(j), sc[j], sh[j]), 0.f));   // round to T like the stored output
            const float f = q.get(j);
            if (f > best[j]) { best[j] = f; arg[j] = kh * 3 + kw; }
          }
        }
      }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.set(j, best[j]);
    uint8_t am[V];
#pragma unroll
    for (int j = 0; j < V; ++j) am[j] = (uint8_t)arg[j];
    if constexpr (V == 8) *reinterpret_cast<uint2*>(argmax + i * V) = *reinterpret_cast<uint2*>(am);
    else *reinterpret_cast<uint32_t*>(argmax + i * V) = *reinterpret_cast<uint32_t*>(am);
    reinterpret_cast<uint4*>(out)[i] = o.raw;
  }
}

// x / d for x * d < 2^32 by one multiply-high (m = ceil(2^32 / d)): the pool kernels decompose a flat index per element,
// and 64-bit divisions by run-time W, H were most of their instructions
struct FastDiv {
  unsigned d, m;
  __device__ __forceinline__ explicit FastDiv(int d_) : d((unsigned)d_), m((unsigned)((0x100000000ull + (unsigned)d_ - 1) / (unsigned)d_)) {}
  __device__ __forceinline__ unsigned div(unsigned x) const { return d == 1 ? x : __umulhi(x, m); }
};

// gradient w.r.t. the post-ReLU stem activation: gather from the <=4 windows covering a pixel
template <typename T>
__global__ __launch_bounds__(256) void k_pool_bwd(int N, int H, int W, int C, const T* __restrict__ dout,
                                                  const uint8_t* __restrict__ argmax, T* __restrict__ dpost) {
  constexpr int V = Vec16<T>::N;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C / V;
  const long total = (long)N * H * W * G;
  // XCD-contiguous index ranges (block b runs on XCD b % 8): the overlapping 3x3 windows of neighbouring rows are
  // then fetched into ONE L2 instead of up to four (225 -> ~150 MB of HBM traffic for the backward)
  const long chunk = (total + 7) / 8, lo = (long)(blockIdx.x & 7) * chunk, hi = lo + chunk < total ? lo + chunk : total;
  const FastDiv dG(G), dW(W), dH(H);                 // (total < 2^31 and the divisor bound are checked on the host)
  for (long i = lo + (long)(blockIdx.x >> 3) * 256 + threadIdx.x; i < hi; i += (long)(gridDim.x >> 3) * 256) {
    const unsigned p0 = dG.div((unsigned)i), p1 = dW.div(p0), p2 = dH.div(p1);
    const int g = (int)((unsigned)i - p0 * (unsigned)G), w = (int)(p0 - p1 * (unsigned)W), h = (int)(p1 - p2 * (unsigned)H), n = (int)p2;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int oh = (h >> 1); oh <= ((h + 1) >> 1); ++oh) {
      if (oh >= Ho) continue;
      const int kh = h - (oh * 2 - 1);
      for (int ow = (w >> 1); ow <= ((w + 1) >> 1); ++ow) {
        if (ow >= Wo) continue;
        const int code = kh * 3 + (w - (ow * 2 - 1));
        const long o = ((((long)n * Ho + oh) * Wo + ow) * G + g);
        Vec16<T> d;
        d.raw = reinterpret_cast<const uint4*>(dout)[o];
        // the V window codes of this group as ONE load (byte loads were the bulk of this kernel's memory ops)
        uint8_t am[V];
        if constexpr (V == 8) *reinterpret_cast<uint2*>(am) = *reinterpret_cast<const uint2*>(argmax + o * V);
        else *reinterpret_cast<uint32_t*>(am) = *reinterpret_cast<const uint32_t*>(argmax + o * V);
#pragma unroll
        for (int j = 0; j < V; ++j) if (am[j] == code) acc[j] += d.get(j);
      }
    }
    Vec16<T> r;
#pragma unroll
    for (int j = 0; j < V; ++j) r.set(j, acc[j]);
    reinterpret_cast<uint4*>(dpost)[i] = r.raw;
  }
}

// Stem backward without the full-resolution gradient in memory: the max-pool gather (above) feeds the BatchNorm/ReLU
// backward directly.  APPLY = false: dz = gather * (scale*y + shift > 0); per-block partial sums (sum dz, sum dz*xhat)
// -> partial [gridDim][2][C] (the layout frx_bn_bwd_finalize reads).  APPLY = true: the same dz, then
// dy = alpha*dz + beta*y + gam -> dy.  Both passes re-gather from the 4x smaller pooled gradient (L2-resident per XCD)
// instead of writing the [N,H,W,C] gradient once and reading it twice.
template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void k_stem_bwd(int N, int H, int W, int C, const T* __restrict__ dout,
                                                  const uint8_t* __restrict__ argmax, const T* __restrict__ y,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  const float* coef, T* __restrict__ dy,
                                                  float* __restrict__ partial, BnTot bt, int tot_R) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][256][V + 1];
  if constexpr (APPLY) {
    if (bt.tot) {
      stage_bwd_consts(bt, C, s_bn_dyn);
      coef = s_bn_dyn;
      __syncthreads();
    }
  }
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C / V;
  // One thread = one 2x2 pixel quad (rows 2a, 2a+1; columns 2b, 2b+1) of one channel group: the four windows (a..a+1,
  // b..b+1) that cover it are loaded ONCE and serve its four pixels (9 of the 16 pixel-window pairs are live), instead
  // of 2.25 window visits per pixel with lane-divergent trip counts.
  const int Hq = (H + 1) >> 1, Wq = (W + 1) >> 1;
  const long total = (long)N * Hq * Wq * G;
  const long chunk = (total + 7) / 8, lo = (long)(blockIdx.x & 7) * chunk, hi = lo + chunk < total ? lo + chunk : total;
  const long first = lo + (long)(blockIdx.x >> 3) * 256 + threadIdx.x, step = (long)(gridDim.x >> 3) * 256;
  // (256 % G == 0, checked on the host: a thread keeps ONE channel group, so its constants stay in registers)
  const int c = (int)(first % G) * V;
  float sc[V], sh[V], mu[V], is[V], al[V], be[V], ga[V], s1[V], s2[V];
  load_consts<V>(scale + c, sc); load_consts<V>(shift + c, sh);
  if constexpr (APPLY) { load_consts<V>(coef + c, al); load_consts<V>(coef + C + c, be); load_consts<V>(coef + 2 * C + c, ga); }
  else { load_consts<V>(mean + c, mu); load_consts<V>(invstd + c, is); fill_consts<V>(0.f, s1); fill_consts<V>(0.f, s2); }
  const FastDiv dG(G), dW(Wq), dH(Hq);               // (the index bound is checked on the host)
  for (long i = first; i < hi; i += step) {
    const unsigned p0 = dG.div((unsigned)i), p1 = dW.div(p0), p2 = dH.div(p1);
    const int g = (int)((unsigned)i - p0 * (unsigned)G), b = (int)(p0 - p1 * (unsigned)Wq), a = (int)(p1 - p2 * (unsigned)Hq), n = (int)p2;
    // the four windows: index 2*dh + dw  <->  (a + dh, b + dw); absent ones (past the pooled grid) read window (a, b)
    // again under a code no pixel matches
    Vec16<T> d[4];
    uint8_t am[4][V];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oh = a + (k >> 1), ow = b + (k & 1);
      const bool ok = oh < Ho && ow < Wo;
      const long o = ((((long)n * Ho + (ok ? oh : a)) * Wo + (ok ? ow : b)) * G + g);
      d[k].raw = reinterpret_cast<const uint4*>(dout)[o];
      if constexpr (V == 8) *reinterpret_cast<uint2*>(am[k]) = *reinterpret_cast<const uint2*>(argmax + o * V);
      else *reinterpret_cast<uint32_t*>(am[k]) = *reinterpret_cast<const uint32_t*>(argmax + o * V);
      if (!ok) {
#pragma unroll
        for (int j = 0; j < V; ++j) am[k][j] = 255;
      }
    }
    Vec16<T> vy[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = 2 * a + (q >> 1), w = 2 * b + (q & 1);
      const bool ok = h < H && w < W;
      vy[q].raw = reinterpret_cast<const uint4*>(y)[((((long)n * H + (ok ? h : 2 * a)) * W + (ok ? w : 2 * b)) * G + g)];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ph = q >> 1, pw = q & 1;                 // pixel (2a + ph, 2b + pw)
      const int h = 2 * a + ph, w = 2 * b + pw;
      if (h >= H || w >= W) continue;                    // (odd H / W: the quad hangs over the edge)
      float acc[V];
      fill_consts<V>(0.f, acc);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dh = k >> 1, dw = k & 1;
        if ((dh && !ph) || (dw && !pw)) continue;        // even rows / columns lie in ONE window row / column
        const int code = (ph + 1 - 2 * dh) * 3 + (pw + 1 - 2 * dw);      // kh = h - (2*oh - 1), kw likewise
#pragma unroll
        for (int j = 0; j < V; ++j) if (am[k][j] == code) acc[j] += d[k].get(j);
      }
      Vec16<T> r;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float yy = vy[q].get(j);
        // (the gathered sum is rounded to T first: what the stand-alone pool backward stored)
        Vec16<T> t1; t1.set(0, acc[j]);
        const float dz = fmaf(yy, sc[j], sh[j]) > 0.f ? t1.get(0) : 0.f;
        if constexpr (APPLY) r.set(j, fmaf(al[j], dz, fmaf(be[j], yy, ga[j])));
        else { s1[j] += dz; s2[j] += dz * (yy - mu[j]) * is[j]; }
      }
      if constexpr (APPLY) reinterpret_cast<uint4*>(dy)[(((long)n * H + h) * W + w) * G + g] = r.raw;
    }
  }
  if constexpr (!APPLY) {
#pragma unroll
    for (int j = 0; j < V; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < G) {      // thread t sums the threads t, t + G, ... (all on t's channel group)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float a = 0.f, b = 0.f;
        for (int k = threadIdx.x; k < 256; k += G) { a += red[0][k][j]; b += red[1][k][j]; }
        if (tot_R > 0) {
          __hip_atomic_fetch_add(partial + ((long)(blockIdx.x % tot_R) * 2 + 0) * C + c + j, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(partial + ((long)(blockIdx.x % tot_R) * 2 + 1) * C + c + j, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          partial[((long)blockIdx.x * 2 + 0) * C + c + j] = a;
          partial[((long)blockIdx.x * 2 + 1) * C + c + j] = b;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- average pool over HW
template <typename T>
__global__ __launch_bounds__(256) void k_avgpool_fwd(int N, int HW, int C, const T* __restrict__ x, T* __restrict__ out) {
  const long total = (long)N * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long n = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += load_as_float<T>(x, (n * HW + p) * C + c);
    out[i] = (T)(s / (float)HW);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_avgpool_bwd(int N, int HW, int C, const T* __restrict__ dpool, T* __restrict__ dx) {
  const long total = (long)N * HW * C;
  const float inv = 1.f / (float)HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long n = i / ((long)HW * C);
    dx[i] = (T)(load_as_float<T>(dpool, n * C + c) * inv);
  }
}

// grid for the RowWalk kernels: every thread gets RU rows per trip (all their loads in flight together);
// more blocks than that only re-read clamped rows
static inline int row_grid(long rows, int C, int V) {
  const int groups = C / V;
  const int rpp = groups < 256 ? 256 / groups : 1;
  long b = (rows + (long)rpp * RU - 1) / ((long)rpp * RU);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// Row sweeps whose blocks first derive their constants from replicated totals run fewer, longer-lived blocks: the
// derivation is paid per block
static inline int tot_row_grid(int grid) { return grid < 512 ? grid : 512; }

static inline int pool_grid(long work_items) {       // multiple of 8: the pool kernels split the index space per XCD
  long b = (work_items + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  b = (b + 7) / 8 * 8;
  return (int)b;
}

// FastDiv's exactness bound for the flat (n, h, w, group) index of the pool kernels
static inline bool pool_index_ok(long N, long H, long W, long G) {
  const long total = N * H * W * G, dmax = W > H ? (W > G ? W : G) : (H > G ? H : G);
  return total > 0 && total * dmax < 0x100000000l;
}

static inline int ew_grid(long work_items) {
  long b = (work_items + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace frx
using namespace frx;

static inline bool frx_groups_ok(int groups) { return groups >= 256 ? groups % 256 == 0 : (groups > 0 && 256 % groups == 0); }
#define FRX_DT_CHECK(dt) FRX_CHECK_ARG((dt) == FRX_F32 || (dt) == FRX_BF16, "unsupported dtype %d", (dt))
#define FRX_VEC(dt) ((dt) == FRX_BF16 ? 8 : 4)

extern "C" int frx_bn_finalize(int device, frx_stream_t stream, const float* partial, int rows, int C,
                               int64_t count, const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift) {
  FRX_CHECK_ARG(partial && gamma && beta && mean && invstd && scale && shift, "bn_finalize: NULL pointer");
  FRX_CHECK_ARG(rows > 0 && C > 0 && count > 0, "bn_finalize: bad sizes");
  FRX_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats come together");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 16)), dim3(FIN_THREADS), 0, (hipStream_t)stream, partial, rows, C,
                     (double)count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_bn_eval_affine(int device, frx_stream_t stream, int C, const float* gamma, const float* beta,
                                  const float* running_mean, const float* running_var, float eps, float* scale,
                                  float* shift) {
  FRX_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_eval_affine: bad args");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_bn_eval_affine, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

static int check_bn_tot(const frx_bn_tot* t, bool forward, const char* who) {
  FRX_CHECK_ARG(t->totals && t->gamma && (forward ? t->beta != nullptr : (t->mean && t->invstd)), "%s: frx_bn_tot has NULL pointers", who);
  FRX_CHECK_ARG(frx_pow2(t->replicas) && t->count > 0.f, "%s: frx_bn_tot needs a power-of-two replica count and count > 0", who);
  return FRX_OK;
}

static int merge_impl(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* y3,
                      const float* s3, const float* b3, const void* idn, const float* sd,
                      const float* bd, void* out, uint8_t* mask, const frx_bn_tot* t3 = nullptr, const frx_bn_tot* td = nullptr) {
  FRX_DT_CHECK(dtype);
  FRX_CHECK_ARG(y3 && (t3 || (s3 && b3)) && idn && out && rows > 0 && C % FRX_VEC(dtype) == 0, "block_merge_fwd: bad args");
  if (t3) { if (int rc = check_bn_tot(t3, true, "block_merge_fwd_tot")) return rc; }
  if (td) { if (int rc = check_bn_tot(td, true, "block_merge_fwd_tot")) return rc; }
  FRX_CHECK_ARG(!t3 || C <= 4096, "block_merge_fwd_tot: C=%d too wide for the LDS tables", C);
  const unsigned dyn = t3 ? (unsigned)((td ? 4 : 2) * C * sizeof(float)) : 0u;
  const BnTot a3 = bn_tot_arg(t3), ad = bn_tot_arg(td);
  FRX_CHECK_ARG(frx_groups_ok(C / FRX_VEC(dtype)), "block_merge_fwd: C=%d must give a power-of-two number of 16-byte groups", C);
  FRX_CHECK_ARG((sd == nullptr) == (bd == nullptr), "block_merge_fwd: sd/bd come together");
  FRX_ENTER(device);
  int grid = row_grid(rows, C, FRX_VEC(dtype));
  if (t3) grid = tot_row_grid(grid);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_merge_fwd<bf16_t>, dim3(grid), dim3(256), dyn, (hipStream_t)stream, (long)rows, C,
                       (const bf16_t*)y3, s3, b3, (const bf16_t*)idn, sd, bd, (bf16_t*)out, mask, a3, ad);
  else
    hipLaunchKernelGGL(k_merge_fwd<float>, dim3(grid), dim3(256), dyn, (hipStream_t)stream, (long)rows, C,
                       (const float*)y3, s3, b3, (const float*)idn, sd, bd, (float*)out, mask, a3, ad);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_block_merge_fwd(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* y3,
                                   const float* s3, const float* b3, const void* idn, const float* sd,
                                   const float* bd, void* out) {
  return merge_impl(device, stream, dtype, rows, C, y3, s3, b3, idn, sd, bd, out, nullptr);
}

extern "C" int frx_block_merge_fwd_mask(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* y3,
                                        const float* s3, const float* b3, const void* idn, const float* sd,
                                        const float* bd, void* out, uint8_t* mask) {
  FRX_CHECK_ARG(mask != nullptr, "block_merge_fwd_mask: mask is NULL");
  return merge_impl(device, stream, dtype, rows, C, y3, s3, b3, idn, sd, bd, out, mask);
}

extern "C" int frx_block_merge_fwd_tot(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* y3,
                                       const frx_bn_tot* bn3, const void* idn, const frx_bn_tot* bnd, void* out, uint8_t* mask) {
  FRX_CHECK_ARG(bn3 != nullptr, "block_merge_fwd_tot: bn3 is NULL");
  return merge_impl(device, stream, dtype, rows, C, y3, nullptr, nullptr, idn, nullptr, nullptr, out, mask, bn3, bnd);
}

extern "C" int frx_bn_bwd_partial_rows(int64_t rows, int C) {
  // one partial row per block of k_bn_bwd_reduce; a block's threads each take RU rows per trip.  The dtype is
  // not known here: 8 channels per 16-byte group (bf16) gives the larger count, which fp32 callers over-allocate.
  const int groups = C >= 8 ? C / 8 : 1;
  const long rpp = groups < 256 ? 256 / groups : 1;
  long b = (rows + rpp * RU - 1) / (rpp * RU);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

static int bwd_reduce_impl(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g,
                           const void* y, const void* out, const float* scale, const float* shift, int relu,
                           const float* mean, const float* invstd, void* dz_out, float* partial, int g_pool_hw, int tot_R) {
  FRX_DT_CHECK(dtype);
  FRX_CHECK_ARG(tot_R == 0 || frx_pow2(tot_R), "bn_bwd_reduce_tot: replicas must be a power of two");
  FRX_CHECK_ARG(g && y && mean && invstd && partial && rows > 0, "bn_bwd_reduce: bad args");
  FRX_CHECK_ARG(g_pool_hw >= 0 && (g_pool_hw == 0 || rows % g_pool_hw == 0), "bn_bwd_reduce: rows=%ld is not a multiple of g_pool_hw=%d", (long)rows, g_pool_hw);
  const int V = FRX_VEC(dtype), groups = C / V;
  FRX_CHECK_ARG(C % V == 0 && (groups >= 256 ? groups % 256 == 0 : 256 % groups == 0),
                "bn_bwd_reduce: C=%d must give a power-of-two number of 16-byte groups", C);
  FRX_CHECK_ARG(out || !relu || (scale && shift), "bn_bwd_reduce: ReLU mask needs out or scale/shift");
  FRX_ENTER(device);
  const int nblk = frx_bn_bwd_partial_rows(rows, C);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_bn_bwd_reduce<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (long)rows, C,
                       (const bf16_t*)g, (const bf16_t*)y, (const bf16_t*)out, scale, shift, relu, mean, invstd,
                       (bf16_t*)dz_out, partial, g_pool_hw, tot_R);
  else
    hipLaunchKernelGGL(k_bn_bwd_reduce<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (long)rows, C,
                       (const float*)g, (const float*)y, (const float*)out, scale, shift, relu, mean, invstd,
                       (float*)dz_out, partial, g_pool_hw, tot_R);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_bn_bwd_reduce(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g,
                                 const void* y, const void* out, const float* scale, const float* shift, int relu,
                                 const float* mean, const float* invstd, void* dz_out, float* partial, int g_pool_hw) {
  return bwd_reduce_impl(device, stream, dtype, rows, C, g, y, out, scale, shift, relu, mean, invstd, dz_out, partial, g_pool_hw, 0);
}
extern "C" int frx_bn_bwd_reduce_tot(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g,
                                     const void* y, const void* out, const float* scale, const float* shift, int relu,
                                     const float* mean, const float* invstd, void* dz_out, float* totals, int replicas,
                                     int g_pool_hw) {
  FRX_CHECK_ARG(replicas > 0, "bn_bwd_reduce_tot: replicas must be positive");
  return bwd_reduce_impl(device, stream, dtype, rows, C, g, y, out, scale, shift, relu, mean, invstd, dz_out, totals, g_pool_hw, replicas);
}

extern "C" int frx_bn_bwd_finalize(int device, frx_stream_t stream, const float* partial, int nblk, int C,
                                   int64_t count, const float* gamma, const float* mean, const float* invstd,
                                   float* dgamma, float* dbeta, float* coef) {
  FRX_CHECK_ARG(partial && gamma && mean && invstd && coef && nblk > 0 && C > 0 && count > 0, "bn_bwd_finalize: bad args");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(cdiv(C, 16)), dim3(FIN_THREADS), 0, (hipStream_t)stream, partial, nblk, C,
                     (double)count, gamma, mean, invstd, dgamma, dbeta, coef);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

static int bwd_apply_impl(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g,
                          const void* y, const void* out, const float* scale, const float* shift, int relu,
                          const float* mean, const float* invstd, const float* coef, void* dy, const frx_bn_tot* bt) {
  FRX_DT_CHECK(dtype);
  FRX_CHECK_ARG(g && y && (bt || (mean && invstd && coef)) && dy && rows > 0 && C % FRX_VEC(dtype) == 0, "bn_bwd_apply: bad args");
  if (bt) { if (int rc = check_bn_tot(bt, false, "bn_bwd_apply_tot")) return rc; }
  FRX_CHECK_ARG(!bt || C <= 4096, "bn_bwd_apply_tot: C=%d too wide for the LDS tables", C);
  const unsigned dyn = bt ? (unsigned)(3 * C * sizeof(float)) : 0u;
  const BnTot ab = bn_tot_arg(bt);
  FRX_CHECK_ARG(frx_groups_ok(C / FRX_VEC(dtype)), "bn_bwd_apply: C=%d must give a power-of-two number of 16-byte groups", C);
  FRX_CHECK_ARG(out || !relu || (scale && shift), "bn_bwd_apply: ReLU mask needs out or scale/shift");
  FRX_ENTER(device);
  int grid = row_grid(rows, C, FRX_VEC(dtype));
  if (bt) grid = tot_row_grid(grid);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_bn_bwd_apply<bf16_t>, dim3(grid), dim3(256), dyn, (hipStream_t)stream, (long)rows, C,
                       (const bf16_t*)g, (const bf16_t*)y, (const bf16_t*)out, scale, shift, relu, mean, invstd, coef,
                       (bf16_t*)dy, ab);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(grid), dim3(256), dyn, (hipStream_t)stream, (long)rows, C,
                       (const float*)g, (const float*)y, (const float*)out, scale, shift, relu, mean, invstd, coef,
                       (float*)dy, ab);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_bn_bwd_apply(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g,
                                const void* y, const void* out, const float* scale, const float* shift, int relu,
                                const float* mean, const float* invstd, const float* coef, void* dy) {
  return bwd_apply_impl(device, stream, dtype, rows, C, g, y, out, scale, shift, relu, mean, invstd, coef, dy, nullptr);
}
extern "C" int frx_bn_bwd_apply_tot(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g,
                                    const void* y, const void* out, const float* scale, const float* shift, int relu,
                                    const frx_bn_tot* bn, void* dy) {
  FRX_CHECK_ARG(bn != nullptr, "bn_bwd_apply_tot: bn is NULL");
  return bwd_apply_impl(device, stream, dtype, rows, C, g, y, out, scale, shift, relu, bn->mean, bn->invstd, nullptr, dy, bn);
}

static int pool_fwd_impl(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C,
                         const void* y, const float* scale, const float* shift, void* out, uint8_t* argmax, const frx_bn_tot* bt) {
  FRX_DT_CHECK(dtype);
  FRX_CHECK_ARG(y && (bt || (scale && shift)) && out && argmax && N > 0 && H > 1 && W > 1 && C > 0 && C % FRX_VEC(dtype) == 0, "stem_pool_fwd: bad args");
  if (bt) { if (int rc = check_bn_tot(bt, true, "stem_pool_fwd_tot")) return rc; }
  const unsigned dyn = bt ? (unsigned)(2 * C * sizeof(float)) : 0u;
  const BnTot ab = bn_tot_arg(bt);
  FRX_ENTER(device);
  const long total = (long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C / FRX_VEC(dtype);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_pool_fwd<bf16_t>, dim3(pool_grid(total)), dim3(256), dyn, (hipStream_t)stream, N, H, W, C,
                       (const bf16_t*)y, scale, shift, (bf16_t*)out, argmax, ab);
  else
    hipLaunchKernelGGL(k_pool_fwd<float>, dim3(pool_grid(total)), dim3(256), dyn, (hipStream_t)stream, N, H, W, C,
                       (const float*)y, scale, shift, (float*)out, argmax, ab);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_stem_pool_fwd(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C,
                                 const void* y, const float* scale, const float* shift, void* out, uint8_t* argmax) {
  return pool_fwd_impl(device, stream, dtype, N, H, W, C, y, scale, shift, out, argmax, nullptr);
}
extern "C" int frx_stem_pool_fwd_tot(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C,
                                     const void* y, const frx_bn_tot* bn, void* out, uint8_t* argmax) {
  FRX_CHECK_ARG(bn != nullptr, "stem_pool_fwd_tot: bn is NULL");
  return pool_fwd_impl(device, stream, dtype, N, H, W, C, y, nullptr, nullptr, out, argmax, bn);
}

extern "C" int frx_stem_pool_bwd(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C,
                                 const void* dout, const uint8_t* argmax, void* dpost) {
  FRX_DT_CHECK(dtype);
  FRX_CHECK_ARG(dout && argmax && dpost && N > 0 && H > 1 && W > 1 && C > 0 && C % FRX_VEC(dtype) == 0, "stem_pool_bwd: bad args");
  FRX_CHECK_ARG(pool_index_ok(N, H, W, C / FRX_VEC(dtype)), "stem_pool_bwd: N*H*W*C/%d must stay below 2^31 / max(W, H, C)", FRX_VEC(dtype));
  FRX_ENTER(device);
  const long total = (long)N * H * W * C / FRX_VEC(dtype);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_pool_bwd<bf16_t>, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, N, H, W, C,
                       (const bf16_t*)dout, argmax, (bf16_t*)dpost);
  else
    hipLaunchKernelGGL(k_pool_bwd<float>, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream, N, H, W, C,
                       (const float*)dout, argmax, (float*)dpost);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

// Fused stem backward (pool gather -> ReLU mask -> BatchNorm backward), two passes around frx_bn_bwd_finalize:
//   frx_stem_bwd_reduce -> partial [frx_stem_bwd_partial_rows()][2][C];  frx_stem_bwd_apply -> dy
extern "C" int frx_stem_bwd_partial_rows(void) { return 1024; }

static int stem_bwd_impl(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                         const uint8_t* argmax, const void* y, const float* scale, const float* shift, const float* mean,
                         const float* invstd, const float* coef, void* dy, float* partial, const frx_bn_tot* bt = nullptr, int tot_R = 0) {
  FRX_DT_CHECK(dtype);
  if (bt) { if (int rc = check_bn_tot(bt, false, "stem_bwd_apply_tot")) return rc; }
  FRX_CHECK_ARG(tot_R == 0 || frx_pow2(tot_R), "stem_bwd_reduce_tot: replicas must be a power of two");
  const unsigned dyn = bt ? (unsigned)(3 * C * sizeof(float)) : 0u;
  const BnTot ab = bn_tot_arg(bt);
  FRX_CHECK_ARG(dout && argmax && y && scale && shift && N > 0 && H > 1 && W > 1 && C > 0 && C % FRX_VEC(dtype) == 0, "stem_bwd: bad args");
  const int G = C / FRX_VEC(dtype);
  FRX_CHECK_ARG(G <= 256 && 256 % G == 0, "stem_bwd: C=%d must give a power-of-two number (<= 256) of 16-byte groups", C);
  FRX_CHECK_ARG(pool_index_ok(N, H, W, G), "stem_bwd: N*H*W*C/%d must stay below 2^31 / max(W, H, C)", FRX_VEC(dtype));
  FRX_ENTER(device);
  const dim3 grid(dy ? pool_grid((long)N * ((H + 1) / 2) * ((W + 1) / 2) * G) : frx_stem_bwd_partial_rows()), blk(256);      // (both multiples of 8)
  const hipStream_t st = (hipStream_t)stream;
  if (dy) {
    if (dtype == FRX_BF16) hipLaunchKernelGGL((k_stem_bwd<bf16_t, true>), grid, blk, dyn, st, N, H, W, C, (const bf16_t*)dout, argmax, (const bf16_t*)y, scale, shift, mean, invstd, coef, (bf16_t*)dy, partial, ab, tot_R);
    else hipLaunchKernelGGL((k_stem_bwd<float, true>), grid, blk, dyn, st, N, H, W, C, (const float*)dout, argmax, (const float*)y, scale, shift, mean, invstd, coef, (float*)dy, partial, ab, tot_R);
  } else {
    if (dtype == FRX_BF16) hipLaunchKernelGGL((k_stem_bwd<bf16_t, false>), grid, blk, dyn, st, N, H, W, C, (const bf16_t*)dout, argmax, (const bf16_t*)y, scale, shift, mean, invstd, coef, (bf16_t*)dy, partial, ab, tot_R);
    else hipLaunchKernelGGL((k_stem_bwd<float, false>), grid, blk, dyn, st, N, H, W, C, (const float*)dout, argmax, (const float*)y, scale, shift, mean, invstd, coef, (float*)dy, partial, ab, tot_R);
  }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_stem_bwd_reduce(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                                   const uint8_t* argmax, const void* y, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, float* partial) {
  FRX_CHECK_ARG(mean && invstd && partial, "stem_bwd_reduce: NULL pointer");
  return stem_bwd_impl(device, stream, dtype, N, H, W, C, dout, argmax, y, scale, shift, mean, invstd, nullptr, nullptr, partial);
}

extern "C" int frx_stem_bwd_apply(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                                  const uint8_t* argmax, const void* y, const float* scale, const float* shift,
                                  const float* coef, void* dy) {
  FRX_CHECK_ARG(coef && dy, "stem_bwd_apply: NULL pointer");
  return stem_bwd_impl(device, stream, dtype, N, H, W, C, dout, argmax, y, scale, shift, nullptr, nullptr, coef, dy, nullptr);
}

extern "C" int frx_stem_bwd_reduce_tot(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                                       const uint8_t* argmax, const void* y, const float* scale, const float* shift,
                                       const float* mean, const float* invstd, float* totals, int replicas) {
  FRX_CHECK_ARG(mean && invstd && totals && replicas > 0, "stem_bwd_reduce_tot: bad args");
  return stem_bwd_impl(device, stream, dtype, N, H, W, C, dout, argmax, y, scale, shift, mean, invstd, nullptr, nullptr, totals, nullptr, replicas);
}

extern "C" int frx_stem_bwd_apply_tot(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                                      const uint8_t* argmax, const void* y, const float* scale, const float* shift,
                                      const frx_bn_tot* bn, void* dy) {
  FRX_CHECK_ARG(bn && dy, "stem_bwd_apply_tot: NULL pointer");
  return stem_bwd_impl(device, stream, dtype, N, H, W, C, dout, argmax, y, scale, shift, nullptr, nullptr, nullptr, dy, nullptr, bn, 0);
}

// ---- the batched closing launches of bn_tot.h: one thread per channel of every listed BatchNorm layer
// forward row  [16] int64: {totals, replicas, C, count, gamma, beta, running_mean | 0, running_var | 0, mean, invstd, scale,
//                           shift, first block, eps (float bits), momentum (float bits), 0}
// backward row [16] int64: {totals, replicas, C, count, gamma, mean, invstd, dgamma | 0, dbeta | 0, coef, 0, 0, first block, 0, 0, 0}
namespace frx {
__device__ __forceinline__ const int64_t* batched_row(const int64_t* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;                 // last row whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 16 + 12] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  return table + lo * 16;
}
__global__ __launch_bounds__(256) void k_bn_finalize_batched(const int64_t* __restrict__ table, int n) {
  const int64_t* row = batched_row(table, n);
  float* tot = (float*)row[0];
  const int R = (int)row[1], C = (int)row[2];
  const int c = ((int)blockIdx.x - (int)row[12]) * 256 + threadIdx.x;
  if (c >= C) return;
  const double count = (double)row[3];
  double s, q;
  bn_tot_sum(tot, R, C, c, s, q);
  float mean, invstd, sc, sh; double var;
  bn_fwd_consts(s, q, 1.0 / count, ((const float*)row[4])[c], ((const float*)row[5])[c], __int_as_float((int)row[13]), mean, invstd, sc, sh, var);
  ((float*)row[8])[c] = mean; ((float*)row[9])[c] = invstd; ((float*)row[10])[c] = sc; ((float*)row[11])[c] = sh;
  if (row[6]) {
    const float momentum = __int_as_float((int)row[14]);
    float* rmean = (float*)row[6]; float* rvar = (float*)row[7];
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
  if (row[15] && c == 0) *(int64_t*)row[15] += 1;                 // BatchNorm2d.num_batches_tracked of this layer
  for (int r = 0; r < 2 * R; ++r) tot[r * C + c] = 0.f;          // ready for the next step's adds
}
__global__ __launch_bounds__(256) void k_bn_bwd_finalize_batched(const int64_t* __restrict__ table, int n) {
  const int64_t* row = batched_row(table, n);
  float* tot = (float*)row[0];
  const int R = (int)row[1], C = (int)row[2];
  const int c = ((int)blockIdx.x - (int)row[12]) * 256 + threadIdx.x;
  if (c >= C) return;
  double a, b;
  bn_tot_sum(tot, R, C, c, a, b);
  if (row[8]) ((float*)row[8])[c] += (float)a;            // dbeta
  if (row[7]) ((float*)row[7])[c] += (float)b;            // dgamma
  float al, be, ga;
  bn_bwd_consts(a, b, 1.0 / (double)row[3], ((const float*)row[4])[c], ((const float*)row[5])[c], ((const float*)row[6])[c], al, be, ga);
  float* coef = (float*)row[9];
  coef[c] = al; coef[C + c] = be; coef[2 * C + c] = ga;
  for (int r = 0; r < 2 * R; ++r) tot[r * C + c] = 0.f;
}
}  // namespace frx

extern "C" int frx_bn_finalize_batched(int device, frx_stream_t stream, int n, const int64_t* table_dev, int total_blocks) {
  FRX_CHECK_ARG(n > 0 && table_dev && total_blocks > 0, "bn_finalize_batched: bad args");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_bn_finalize_batched, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
extern "C" int frx_bn_bwd_finalize_batched(int device, frx_stream_t stream, int n, const int64_t* table_dev, int total_blocks) {
  FRX_CHECK_ARG(n > 0 && table_dev && total_blocks > 0, "bn_bwd_finalize_batched: bad args");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_bn_bwd_finalize_batched, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_avgpool_fwd(int device, frx_stream_t stream, int dtype, int N, int HW, int C, const void* x, void* out) {
  FRX_DT_CHECK(dtype);
  FRX_CHECK_ARG(x && out && N > 0 && HW > 0 && C > 0, "avgpool_fwd: bad args");
  FRX_ENTER(device);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_avgpool_fwd<bf16_t>, dim3(ew_grid((long)N * C)), dim3(256), 0, (hipStream_t)stream, N, HW, C,
                       (const bf16_t*)x, (bf16_t*)out);
  else
    hipLaunchKernelGGL(k_avgpool_fwd<float>, dim3(ew_grid((long)N * C)), dim3(256), 0, (hipStream_t)stream, N, HW, C,
                       (const float*)x, (float*)out);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_avgpool_bwd(int device, frx_stream_t stream, int dtype, int N, int HW, int C, const void* dpool, void* dx) {
  FRX_DT_CHECK(dtype);
  FRX_CHECK_ARG(dpool && dx && N > 0 && HW > 0 && C > 0, "avgpool_bwd: bad args");
  FRX_ENTER(device);
  if (dtype == FRX_BF16)
    hipLaunchKernelGGL(k_avgpool_bwd<bf16_t>, dim3(ew_grid((long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, N, HW, C,
                       (const bf16_t*)dpool, (bf16_t*)dx);
  else
    hipLaunchKernelGGL(k_avgpool_bwd<float>, dim3(ew_grid((long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, N, HW, C,
                       (const float*)dpool, (float*)dx);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
