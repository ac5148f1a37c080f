// Margin-softmax head: normalise -> cosine GEMM (exact-fp32 MFMA) -> margin / CE / top-k row
// sweep, and the analytic backward.  One code path for ArcFace / CosFace / SphereFace /
// CurricularFace (reference: main_code/utils/criterion.py:260-301, 162-197, 57-107, 527-587;
// CE model_utils.py:556,179; top-k metrics.py:3-16).
//
// Numerics: logits are cosines x 64 and the parity bar is 1e-3, which bf16 operands miss by
// 35x: the GEMMs run on the f32-input MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma
// chain).  The split-bf16 scheme of SURVEY H2 (three bf16 MFMA passes on hi / lo halves of each operand) was built
// twice -- 64 x 64 tiles in round 3, 128 x 128 eight-wave tiles in round 4 -- correct on every golden and no faster
// (these products are bound by staging and latency, not by the matrix pipe): removed.
#include "frx_common.h"
#include <type_traits>

namespace frx {

// ------------------------------------------------------------------------------------------
// Generic fp32 GEMM on v_mfma_f32_32x32x2_f32.  64x64 tile, BK=16, 4 waves (2x2 of 32x32).
// Operand layouts cover every product the head needs without materialising a transpose.
// ------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long lda, ldb, ldc;
  int a_mcontig;          // 0: A(m,k) = A[m*lda + k]     1: A(m,k) = A[k*lda + m]
  int b_ncontig;          // 0: B(k,n) = B[n*ldb + k]     1: B(k,n) = B[k*ldb + n]
  const float* a_kscale;  // optional [K]: A(m,k) *= a_kscale[k]
  const float* row_scale; // optional [M] epilogue
  const float* col_scale; // optional [N] epilogue
  int atomic_out;         // 1: atomicAdd into C (split-K; C pre-zeroed)  0: plain store
  int ksplit_len;         // K range handled per blockIdx.z (multiple of 16)
};

constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = 68;

template <bool MC>  // MC: the non-K dimension is contiguous in memory
__device__ __forceinline__ void gemm_load_tile(const float* __restrict__ P, long ld, int x0, int X,
                                               int k0, int kend, const float* __restrict__ kscale,
                                               float v[4], int& xi, int& ki) {
  const int t = threadIdx.x;
  if (MC) {  // element (x, k) at P[k*ld + x]; thread covers 4 consecutive x at one k
    ki = t >> 4;
    xi = (t & 15) * 4;
    const int k = k0 + ki, x = x0 + xi;
    const bool vec = (k < kend) && (x + 3 < X) && ((ld & 3) == 0) && ((((size_t)P) & 15) == 0);
    if (vec) {
      const float4 q = *reinterpret_cast<const float4*>(P + (long)k * ld + x);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (k < kend && x + j < X) ? P[(long)k * ld + x + j] : 0.f;
    }
    if (kscale) {
      const float sc = (k < kend) ? kscale[k] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= sc;
    }
  } else {   // element (x, k) at P[x*ld + k]; thread covers 4 consecutive k at one x
    xi = t >> 2;
    ki = (t & 3) * 4;
    const int k = k0 + ki, x = x0 + xi;
    const bool vec = (x < X) && (k + 3 < kend) && ((ld & 3) == 0) && ((((size_t)P) & 15) == 0);
    if (vec) {
      const float4 q = *reinterpret_cast<const float4*>(P + (long)x * ld + k);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (x < X && k + j < kend) ? P[(long)x * ld + k + j] : 0.f;
    }
    if (kscale) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= (k + j < kend) ? kscale[k + j] : 0.f;
    }
  }
}

template <bool MC>
__device__ __forceinline__ void gemm_store_tile(float (*S)[GLD], const float v[4], int xi, int ki) {
  if (MC) {
    *reinterpret_cast<float4*>(&S[ki][xi]) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) S[ki + j][xi] = v[j];
  }
}

// the K loop of one 64 x 64 tile: two LDS stages, the global loads of K-step k+1 issued before the MFMAs of step k and stored to
// the other stage after them -- one barrier per step, the loads' latency under the 8 MFMAs (the single-stage load -> barrier
// -> store -> barrier -> MFMA loop left it exposed: 48-75 TFLOP/s, now see DESIGN.md section 8)
template <bool AMC, bool BNC>
__device__ __forceinline__ void gemm_f32_tile(const GemmArgs& g, int m0, int n0, int kbeg, int kend, float (*As)[GBK][GLD],
                                              float (*Bs)[GBK][GLD], f32x16& acc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float va[4], vb[4];
  int ax, ak, bx, bk;
  gemm_load_tile<AMC>(g.A, g.lda, m0, g.M, kbeg, kend, g.a_kscale, va, ax, ak);
  gemm_load_tile<BNC>(g.B, g.ldb, n0, g.N, kbeg, kend, nullptr, vb, bx, bk);
  gemm_store_tile<AMC>(As[0], va, ax, ak);
  gemm_store_tile<BNC>(Bs[0], vb, bx, bk);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    const bool more = k0 + GBK < kend;
    if (more) {
      gemm_load_tile<AMC>(g.A, g.lda, m0, g.M, k0 + GBK, kend, g.a_kscale, va, ax, ak);
      gemm_load_tile<BNC>(g.B, g.ldb, n0, g.N, k0 + GBK, kend, nullptr, vb, bx, bk);
    }
#pragma unroll
    for (int kk = 0; kk < GBK / 2; ++kk) {
      const float a = As[buf][2 * kk + lh][wm * 32 + li];
      const float b = Bs[buf][2 * kk + lh][wn * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (more) {
      gemm_store_tile<AMC>(As[buf ^ 1], va, ax, ak);
      gemm_store_tile<BNC>(Bs[buf ^ 1], vb, bx, bk);
    }
    __syncthreads();
    buf ^= 1;
  }
}

template <bool AMC, bool BNC>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][GBK][GLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][GBK][GLD];
  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const int kbeg = blockIdx.z * g.ksplit_len;
  const int kend = min(g.K, kbeg + g.ksplit_len);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
  gemm_f32_tile<AMC, BNC>(g, m0, n0, kbeg, kend, As, Bs, acc);
  // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int n = n0 + wn * 32 + li;
  if (n >= g.N) return;
  const float cs = g.col_scale ? g.col_scale[n] : 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < g.M) {
      float v = acc[r] * cs;
      if (g.row_scale) v *= g.row_scale[m];
      float* dst = g.C + (long)m * g.ldc + n;
      if (g.atomic_out) atomicAdd(dst, v); else *dst = v;
    }
  }
}

static int launch_gemm(hipStream_t st, GemmArgs g, int ksplit) {
  if (ksplit < 1) ksplit = 1;
  int len = (int)round_up((size_t)cdiv(g.K, ksplit), GBK);
  g.ksplit_len = len;
  ksplit = cdiv(g.K, len);
  g.atomic_out = ksplit > 1 ? 1 : g.atomic_out;
  dim3 grid(cdiv(g.N, GBN), cdiv(g.M, GBM), ksplit), block(256);
  if (g.a_mcontig && g.b_ncontig) hipLaunchKernelGGL((k_gemm_f32<true, true>), grid, block, 0, st, g);
  else if (g.a_mcontig && !g.b_ncontig) hipLaunchKernelGGL((k_gemm_f32<true, false>), grid, block, 0, st, g);
  else if (!g.a_mcontig && g.b_ncontig) hipLaunchKernelGGL((k_gemm_f32<false, true>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((k_gemm_f32<false, false>), grid, block, 0, st, g);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

// ------------------------------------------------------------------------------------------
// Norms
// ------------------------------------------------------------------------------------------
// one wave per row of a [R, D] matrix: nrm = ||row||, inv = 1/max(nrm, 1e-12)   (F.normalize eps)
__global__ __launch_bounds__(256) void k_row_norms(const float* __restrict__ a, int R, int D,
                                                   float* __restrict__ inv, float* __restrict__ nrm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* p = a + (long)row * D;
  float s = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 q = *reinterpret_cast<const float4*>(p + d);
    s += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float n = sqrtf(s);
    if (nrm) nrm[row] = n;
    inv[row] = 1.f / fmaxf(n, 1e-12f);
  }
}

// column norms of a [D, C] matrix (CosFace / CurricularFace `kernel`).  64 columns x 4 row-lanes per block, 8
// independent loads per lane in flight (one thread walking D rows of a column alone is D dependent round trips:
// 206 us for the 174 MB of an 85 000-class head, ~6x the HBM time).
constexpr int COLB = 64, COLR = 4, COLU = 8;
__global__ __launch_bounds__(256) void k_col_norms(const float* __restrict__ a, int D, int C,
                                                   float* __restrict__ inv) {
  __shared__ float red[COLR][COLB];
  const int cl = threadIdx.x % COLB, rl = threadIdx.x / COLB;
  const int c = blockIdx.x * COLB + cl;
  float s = 0.f;
  if (c < C) {
    for (int d0 = rl; d0 < D; d0 += COLR * COLU) {
      float v[COLU];
#pragma unroll
      for (int u = 0; u < COLU; ++u) { const int d = d0 + COLR * u; v[u] = d < D ? a[(long)d * C + c] : 0.f; }
#pragma unroll
      for (int u = 0; u < COLU; ++u) s += v[u] * v[u];
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < COLR; ++i) t += red[i][cl];
    inv[c] = 1.f / fmaxf(sqrtf(t), 1e-12f);
  }
}

// ------------------------------------------------------------------------------------------
// Per-element margin maths shared by forward, loss and backward sweeps
// ------------------------------------------------------------------------------------------
struct HeadConst {
  int kind;
  float s, cos_m, sin_m, th, mm, m, lamb;
  float p0, p1, p2, p3;   // per-kind parameters (frx_head_desc::p)
  int flags;
  int c0;                 // class-sharded head: global index of this shard's first class (labels are global)
};

struct RowCtx {   // per-row values
  float xnorm;    // SPHERE
  float t;        // CURR: EMA value (already updated)
  float ty;       // CURR / MV: clamped target cosine
  float cm;       // CURR / MV: the row's mask threshold (cos(theta_y + m), or ty - m for MV 'am')
  float p;        // ADA: margin scaler; ELASTIC: sampled margin; MAG: adaptive margin
  float aux;      // MAG: d(margin)/d||x||  (slope, or 0 where the norm clamp is active)
};

constexpr float kPi = 3.14159265358979323846f;
constexpr bool kind_is_mv(int k) { return k == FRX_MV_AM || k == FRX_MV_ARC; }
constexpr bool kind_eps7(int k) { return kind_is_mv(k) || k == FRX_ELASTIC_ARC || k == FRX_ELASTIC_COS || k == FRX_MAG || k == FRX_VPL; }

template <int KIND>
__device__ __forceinline__ float head_clamp(float c) {
  if (KIND == FRX_ARC) return c;
  if (KIND == FRX_COS) return fminf(fmaxf(c, -1.f + 1e-4f), 1.f - 1e-4f);
  if (KIND == FRX_ADA) return fminf(fmaxf(c, -1.f + 1e-3f), 1.f - 1e-3f);          // criterion.py:866
  if (kind_eps7(KIND)) return fminf(fmaxf(c, -1.f + 1e-7f), 1.f - 1e-7f);          // :414, :724, :997, :1108, :1261
  return fminf(fmaxf(c, -1.f), 1.f);
}
template <int KIND>
__device__ __forceinline__ bool head_pass(float c) {  // torch clamp backward mask (inclusive)
  if (KIND == FRX_ARC) return true;
  if (KIND == FRX_COS) return c >= -1.f + 1e-4f && c <= 1.f - 1e-4f;
  if (KIND == FRX_ADA) return c >= -1.f + 1e-3f && c <= 1.f - 1e-3f;
  if (kind_eps7(KIND)) return c >= -1.f + 1e-7f && c <= 1.f - 1e-7f;
  return c >= -1.f && c <= 1.f;
}
// pre-margin output ("cos_s", first element of the reference's output list)
template <int KIND>
__device__ __forceinline__ float head_cos_s(float cc, const HeadConst& h, const RowCtx& r) {
  return KIND == FRX_SPHERE ? cc * r.xnorm : cc * h.s;
}
// logit z and dz/dc (w.r.t. the clamped cosine), cc = clamped cosine
template <int KIND>
__device__ __forceinline__ void head_z(float cc, bool target, const HeadConst& h, const RowCtx& r,
                                       float& z, float& dzdc, float& u /*SPHERE: z/||x||*/) {
  u = 0.f;
  if (KIND == FRX_ARC) {
    if (target) {
      const float q = 1.f - cc * cc;
      const float sine = sqrtf(fminf(fmaxf(q, 1e-9f), 1.f));
      const bool easy = (h.flags & 1) != 0;                    // easy_margin (criterion.py:284-285): margin only where cos > 0
      if (easy ? cc > 0.f : cc > h.th) {
        z = (cc * h.cos_m - sine * h.sin_m) * h.s;
        const bool inside = q >= 1e-9f && q <= 1.f;
        dzdc = h.s * (h.cos_m + (inside ? h.sin_m * cc / sine : 0.f));
      } else {
        z = (easy ? cc : cc - h.mm) * h.s;
        dzdc = h.s;
      }
    } else {
      z = cc * h.s;
      dzdc = h.s;
    }
  } else if (KIND == FRX_COS) {
    z = (target ? cc - h.m : cc) * h.s;
    dzdc = h.s;
  } else if (KIND == FRX_SPHERE) {
    if (target) {
      // cos(m theta) as the Chebyshev polynomial T_m(c) (criterion.py:40-47), its derivative m U_{m-1}(c); m = 1..5
      const int mi = (int)(h.m + 0.5f);
      const float c2 = cc * cc;
      float tm, dtm;
      if (mi == 1) { tm = cc; dtm = 1.f; }
      else if (mi == 2) { tm = 2.f * c2 - 1.f; dtm = 4.f * cc; }
      else if (mi == 3) { tm = (4.f * c2 - 3.f) * cc; dtm = 12.f * c2 - 3.f; }
      else if (mi == 4) { tm = (8.f * c2 - 8.f) * c2 + 1.f; dtm = (32.f * c2 - 16.f) * cc; }
      else { tm = ((16.f * c2 - 20.f) * c2 + 5.f) * cc; dtm = (80.f * c2 - 60.f) * c2 + 5.f; }
      const float theta = acosf(cc);
      const float k = floorf((float)mi * theta / 3.14159265358979323846f);
      const float sign = (((int)k) & 1) ? -1.f : 1.f;
      const float phi = sign * tm - 2.f * k;
      u = (phi - cc) / (1.f + h.lamb) + cc;
      dzdc = (1.f + (sign * dtm - 1.f) / (1.f + h.lamb)) * r.xnorm;
    } else {
      u = cc;
      dzdc = r.xnorm;
    }
    z = u * r.xnorm;
  } else if (kind_is_mv(KIND)) {   // criterion.py:420-441
    if (target) {
      if (KIND == FRX_MV_AM) {
        z = (r.ty > h.m ? r.ty - h.m : r.ty) * h.s;
        dzdc = h.s;
      } else if (r.ty > 0.f) {
        z = r.cm * h.s;
        dzdc = h.s * (h.cos_m + h.sin_m * r.ty / sqrtf(1.f - r.ty * r.ty + 1e-9f));
      } else {
        z = r.ty * h.s;
        dzdc = h.s;
      }
    } else if (cc > r.cm) {           // mis-classified vector: re-weighted
      z = (h.p0 * cc + (h.p0 - 1.f)) * h.s;
      dzdc = h.s * h.p0;
    } else {
      z = cc * h.s;
      dzdc = h.s;
    }
  } else if (KIND == FRX_ADA) {       // criterion.py:886-899; non-target entries pass through acos/cos unchanged
    if (target) {
      const float theta = acosf(cc);
      const float raw = theta - h.m * r.p;
      const float tm = fminf(fmaxf(raw, 1e-3f), kPi - 1e-3f);
      z = (cosf(tm) - (h.m + h.m * r.p)) * h.s;
      dzdc = (raw >= 1e-3f && raw <= kPi - 1e-3f) ? h.s * sinf(tm) * rsqrtf(1.f - cc * cc) : 0.f;
    } else {
      z = cc * h.s;
      dzdc = h.s;
    }
  } else if (KIND == FRX_ELASTIC_ARC) {   // criterion.py:1125-1131
    if (target) {
      const float raw = acosf(cc) + r.p;
      const float tm = fminf(fmaxf(raw, 0.f), kPi);
      z = cosf(tm) * h.s;
      dzdc = (raw >= 0.f && raw <= kPi) ? h.s * sinf(tm) * rsqrtf(1.f - cc * cc) : 0.f;
    } else {
      z = cc * h.s;
      dzdc = h.s;
    }
  } else if (KIND == FRX_ELASTIC_COS) {   // criterion.py:1013-1014
    z = (target ? cc - r.p : cc) * h.s;
    dzdc = h.s;
  } else if (KIND == FRX_VPL) {           // criterion.py:727-739, on the blended cosine (frx_head_vpl_prepare)
    if (target) {
      const float sine = sqrtf(1.f - cc * cc + 1e-9f);
      const bool on = (h.flags & 1) ? cc > 0.f : cc > h.th;
      if (on) {
        z = (cc * h.cos_m - sine * h.sin_m) * h.s;
        dzdc = h.s * (h.cos_m + h.sin_m * cc / sine);
      } else {
        z = ((h.flags & 1) ? cc : cc - h.mm) * h.s;
        dzdc = h.s;
      }
    } else {
      z = cc * h.s;
      dzdc = h.s;
    }
  } else if (KIND == FRX_MAG) {           // criterion.py:1264-1284; u = dz/d(margin) feeds the norm gradient
    if (target) {
      const float cm_ = cosf(r.p), sm_ = sinf(r.p);
      const float sinth = sqrtf(1.f - cc * cc + 1e-9f);
      const float ctm = cc * cm_ - sinth * sm_;
      bool margin_on;
      float zoff = cc, uoff = 0.f;
      if (h.flags & 1) {
        margin_on = cc > 0.f;
      } else {
        const float cpm = cosf(kPi - r.p), spm = sinf(kPi - r.p);
        margin_on = cc > cpm;
        zoff = cc - spm * r.p;
        uoff = cpm * r.p - spm;
      }
      if (margin_on) {
        z = ctm * h.s;
        dzdc = h.s * (cm_ + sm_ * cc / sinth);
        u = h.s * (-cc * sm_ - sinth * cm_);
      } else {
        z = zoff * h.s;
        dzdc = h.s;
        u = h.s * uoff;
      }
    } else {
      z = cc * h.s;
      dzdc = h.s;
    }
  } else {  // CURR
    if (target) {
      if (r.ty > h.th) {
        z = r.cm * h.s;
        dzdc = h.s * (h.cos_m + h.sin_m * r.ty / sqrtf(1.f - r.ty * r.ty));
      } else {
        z = (r.ty - h.mm) * h.s;
        dzdc = h.s;
      }
    } else if (cc > r.cm) {
      z = cc * (r.t + cc) * h.s;
      dzdc = h.s * (r.t + 2.f * cc);
    } else {
      z = cc * h.s;
      dzdc = h.s;
    }
  }
}

// A label outside [0, C) (a class-count mismatch between the dataset folders and num_classes) must not become an
// out-of-bounds access: every kernel indexes with the label clamped into range, and the row's loss is poisoned with
// NaN so the mistake shows at the caller's next loss read (the reference's CrossEntropyLoss raises there).
// Class-sharded head (flags bit 3): the desc describes columns [c0, c0 + C) of a wider head; labels stay global, and a
// label outside the shard is simply a row whose target lives on another rank (owned = false, not an error).
__device__ __forceinline__ int safe_label(int64_t y, int C, bool& bad) {
  bad = y < 0 || y >= (int64_t)C;
  return bad ? 0 : (int)y;
}
__device__ __forceinline__ int shard_label(int64_t y, const HeadConst& h, int C, bool& owned, bool& bad) {
  const int64_t yl = y - (int64_t)h.c0;
  owned = yl >= 0 && yl < (int64_t)C;
  bad = !owned && !(h.flags & 8);
  return owned ? (int)yl : -1;          // -1 never equals a column index
}

// target cosine per row (clamped as the head clamps), and its sum over the batch
template <int KIND>
__global__ __launch_bounds__(256) void k_head_ty(HeadConst h, const float* __restrict__ cbuf, int N, long ldc, int C,
                                                 const int64_t* __restrict__ labels,
                                                 float* __restrict__ ty, float* __restrict__ ty_sum) {
  __shared__ float sh[4];
  float part = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    bool bad, owned;
    const int y = shard_label(labels[n], h, C, owned, bad);
    // (a row whose target sits in another shard contributes 0: the caller's SUM all-reduce assembles the global vector)
    const float v = bad ? NAN : (owned ? head_clamp<KIND>(cbuf[(long)n * ldc + y]) : 0.f);
    ty[n] = v;
    part += v;
  }
  const float tot = block_sum256(part, sh);
  if (threadIdx.x == 0 && ty_sum) *ty_sum = tot;
}

// CurricularFace EMA: t = momentum * mean(t_y) + (1 - momentum) * t   (criterion.py:572)
__global__ void k_curr_t_update(float* t, const float* ty_sum, float inv_count, float momentum) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *t = (*ty_sum * inv_count) * momentum + (1.f - momentum) * (*t);
}

__device__ __forceinline__ RowCtx make_row_ctx(int kind, int n, const HeadConst& h, const float* xnorm,
                                               const float* ty, const float* state_t, const float* rowp) {
  RowCtx r;
  r.xnorm = xnorm[n];
  r.t = 0.f; r.ty = 0.f; r.cm = 0.f; r.p = 0.f; r.aux = 0.f;
  if (kind == FRX_CURR) {
    r.t = *state_t;
    r.ty = ty[n];
    r.cm = r.ty * h.cos_m - sqrtf(1.f - r.ty * r.ty) * h.sin_m;
  } else if (kind == FRX_MV_AM) {
    r.ty = ty[n];
    r.cm = r.ty - h.m;                                                           // criterion.py:424
  } else if (kind == FRX_MV_ARC) {
    r.ty = ty[n];
    r.cm = r.ty * h.cos_m - sqrtf(1.f - r.ty * r.ty + 1e-9f) * h.sin_m;           // :427-428
  } else if (kind == FRX_ADA || kind == FRX_ELASTIC_ARC || kind == FRX_ELASTIC_COS) {
    r.p = rowp[n];
  } else if (kind == FRX_MAG) {
    r.p = rowp[n];
    r.aux = (r.xnorm >= h.p2 && r.xnorm <= h.p3) ? (h.p1 - h.p0) / (h.p3 - h.p2) : 0.f;
  }
  return r;
}

// Per-row parameters that depend on the batch of feature norms (one block; N is a batch size).
//   ADA: margin scaler from the EMA of the batch mean / unbiased std of the clamped norms (criterion.py:869-880)
//   MAG: clamped norm, adaptive margin, loss_g (criterion.py:1229-1249)
template <int KIND>
__global__ __launch_bounds__(256) void k_head_rowparam(HeadConst h, const float* __restrict__ xnorm, int N,
                                                       float* __restrict__ state, float* __restrict__ rowp,
                                                       float* __restrict__ xn_out, float* __restrict__ lossg) {
  __shared__ float sh[4];
  if (KIND == FRX_ADA) {
    float part = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) part += fminf(fmaxf(xnorm[n], 0.001f), 100.f);
    const float mean = block_sum256(part, sh) / (float)N;
    __syncthreads();
    part = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
      const float dlt = fminf(fmaxf(xnorm[n], 0.001f), 100.f) - mean;
      part += dlt * dlt;
    }
    const float sd = sqrtf(block_sum256(part, sh) / (float)(N - 1));
    const float bm = mean * h.p1 + (1.f - h.p1) * state[0];
    const float bs = sd * h.p1 + (1.f - h.p1) * state[1];
    __syncthreads();                       // every thread has read the old state
    if (threadIdx.x == 0) { state[0] = bm; state[1] = bs; }
    for (int n = threadIdx.x; n < N; n += 256) {
      const float safe = fminf(fmaxf(xnorm[n], 0.001f), 100.f);
      rowp[n] = fminf(fmaxf((safe - bm) / (bs + 1e-3f) * h.p0, -1.f), 1.f);
    }
  } else {  // MAG
    float part = 0.f;
    const float slope = (h.p1 - h.p0) / (h.p3 - h.p2);
    for (int n = threadIdx.x; n < N; n += 256) {
      const float xn = fminf(fmaxf(xnorm[n], h.p2), h.p3);
      xn_out[n] = xn;
      rowp[n] = slope * (xn - h.p2) + h.p0;
      part += 1.f / (h.p3 * h.p3) * xn + 1.f / xn;
    }
    const float tot = block_sum256(part, sh);
    if (threadIdx.x == 0) *lossg = tot / (float)N;
  }
}

// ---- VPL-ArcFace memory (criterion.py:699-722) ----
// mem[cls] = mean of the RAW features of the batch rows labelled cls, life[cls] = delta.  One block per batch row;
// the first row of each label does the work.
__global__ __launch_bounds__(256) void k_vpl_mem_update(const float* __restrict__ x, const int64_t* __restrict__ labels,
                                                        int N, int D, int C, float delta, float* __restrict__ mem,
                                                        float* __restrict__ life) {
  __shared__ int first, cnt;
  const int n = blockIdx.x;
  const int64_t y = labels[n];
  if (y < 0 || y >= (int64_t)C) return;        // (block-uniform) never WRITE through a bad label; the loss goes NaN
  if (threadIdx.x == 0) { first = N; cnt = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256)
    if (labels[i] == y) { atomicMin(&first, i); atomicAdd(&cnt, 1); }
  __syncthreads();
  if (first != n) return;
  for (int dd = threadIdx.x; dd < D; dd += 256) {
    float acc = 0.f;
    for (int i = n; i < N; ++i)
      if (labels[i] == y) acc += x[(long)i * D + dd];
    mem[(long)y * D + dd] = acc / (float)cnt;
  }
  if (threadIdx.x == 0) life[y] = delta;
}

__global__ __launch_bounds__(256) void k_vpl_life_decay(float* __restrict__ life, int C) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < C) life[j] -= 1.f;
}

// c = (1 - a*lamda) * cos_w + a*lamda * (target ? 1 : cos_mem),  a = life > 0   (in place over cos_w)
__global__ __launch_bounds__(256) void k_vpl_blend(float* __restrict__ cbuf, const float* __restrict__ cmem,
                                                   const float* __restrict__ life, const int64_t* __restrict__ labels,
                                                   int C, long ldc, float lamda) {
  const int n = blockIdx.x;
  const int y = (int)labels[n];
  for (int j = threadIdx.x; j < C; j += 256) {
    const float al = (life[j] > 0.f ? 1.f : 0.f) * lamda;
    const long i = (long)n * ldc + j;
    cbuf[i] = (1.f - al) * cbuf[i] + al * (j == y ? 1.f : cmem[i]);
  }
}

// backward of the blend: g -> (g * (1 - a*lamda) for the weight path, g * a*lamda (non-target) for the memory path)
__global__ __launch_bounds__(256) void k_vpl_split(float* __restrict__ gbuf, float* __restrict__ gmem,
                                                   const float* __restrict__ life, const int64_t* __restrict__ labels,
                                                   int C, int Cpad, float lamda) {
  const int n = blockIdx.x;
  const int y = (int)labels[n];
  for (int j = threadIdx.x; j < Cpad; j += 256) {
    const long i = (long)n * Cpad + j;
    float gw = 0.f, gm = 0.f;
    if (j < C) {
      const float al = (life[j] > 0.f ? 1.f : 0.f) * lamda;
      const float g = gbuf[i];
      gw = g * (1.f - al);
      gm = j == y ? 0.f : g * al;
    }
    gbuf[i] = gw;
    gmem[i] = gm;
  }
}

// Forward row sweep: one 256-thread block per sample.
template <int KIND>
__global__ __launch_bounds__(256) void k_head_rows(HeadConst h, const float* __restrict__ cbuf, int C,
                                                   long ldc, const int64_t* __restrict__ labels,
                                                   const float* __restrict__ xnorm,
                                                   const float* __restrict__ ty,
                                                   const float* __restrict__ state_t,
                                                   const float* __restrict__ rowp,
                                                   float* __restrict__ cos_s_out,
                                                   float* __restrict__ logits_out,
                                                   float* __restrict__ lse_out,
                                                   float* __restrict__ rowloss,
                                                   int32_t* __restrict__ rowrank,
                                                   float* __restrict__ part /* shard mode: [3][N] max, sum-exp, rank */) {
  __shared__ float sh[4];
  __shared__ int shi[4];
  if (KIND == FRX_SPHERE && (h.flags & 4)) h.lamb = *state_t;
  const int n = blockIdx.x;
  bool bad_label, owned;
  const int y = shard_label(labels[n], h, C, owned, bad_label);
  const float* crow = cbuf + (long)n * ldc;
  const RowCtx r = make_row_ctx(KIND, n, h, xnorm, ty, state_t, rowp);
  // the clamped target cosine: k_head_ty left it in ty[] (in shard mode the caller put the all-reduced vector there,
  // so every shard sees the row's target cosine although only one of them holds the column)
  const float cy = ty[n];
  const float cos_s_y = head_cos_s<KIND>(cy, h, r);
  float zy, dummy, u;
  head_z<KIND>(cy, true, h, r, zy, dummy, u);

  // four consecutive classes per thread and trip (16-byte loads; rows are Cpad-strided, Cpad % 64 == 0), HU trips' loads
  // issued before the first use: one block walks a whole row, so its bytes in flight are what bounds it (at C = 85 000
  // the one-load-per-trip form ran at 3 GB/s per block: 113 us for 128 rows)
  constexpr int HU = 4;
  float zmax = -INFINITY;
  int rank = 0;
  for (int j0 = threadIdx.x * 4; j0 < C; j0 += 1024 * HU) {
    float4 c4[HU];
#pragma unroll
    for (int t = 0; t < HU; ++t) c4[t] = *reinterpret_cast<const float4*>(crow + (j0 + 1024 * t < C ? j0 + 1024 * t : j0));
#pragma unroll
    for (int t = 0; t < HU; ++t) {
      const float cv[4] = {c4[t].x, c4[t].y, c4[t].z, c4[t].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + 1024 * t + e;
        if (j < C) {
          const float cc = head_clamp<KIND>(cv[e]);
          float z, d;
          head_z<KIND>(cc, j == y, h, r, z, d, u);
          const float cs = head_cos_s<KIND>(cc, h, r);
          zmax = fmaxf(zmax, z);
          rank += (cs > cos_s_y) ? 1 : 0;
          if (cos_s_out) cos_s_out[(long)n * C + j] = cs;
          if (logits_out) logits_out[(long)n * C + j] = z;
        }
      }
    }
  }
  zmax = block_max256(zmax, sh);
  float se = 0.f;
  for (int j0 = threadIdx.x * 4; j0 < C; j0 += 1024 * HU) {
    float4 c4[HU];
#pragma unroll
    for (int t = 0; t < HU; ++t) c4[t] = *reinterpret_cast<const float4*>(crow + (j0 + 1024 * t < C ? j0 + 1024 * t : j0));
#pragma unroll
    for (int t = 0; t < HU; ++t) {
      const float cv[4] = {c4[t].x, c4[t].y, c4[t].z, c4[t].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + 1024 * t + e;
        if (j < C) {
          const float cc = head_clamp<KIND>(cv[e]);
          float z, d;
          head_z<KIND>(cc, j == y, h, r, z, d, u);
          se += expf(z - zmax);
        }
      }
    }
  }
  se = block_sum256(se, sh);
  rank = wave_sum_i(rank);
  if ((threadIdx.x & 63) == 0) shi[threadIdx.x >> 6] = rank;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int rk = shi[0] + shi[1] + shi[2] + shi[3];
    if (part) {                         // partial softmax statistics of this shard's columns
      const int N = gridDim.x;
      part[n] = zmax; part[N + n] = se; part[2 * N + n] = (float)rk;
    } else {
      const float lse = zmax + logf(se);
      if (lse_out) lse_out[n] = lse;
      rowloss[n] = bad_label ? NAN : lse - zy;
      rowrank[n] = rk;
    }
  }
}

// shard mode, after the partial statistics were combined across shards: part_sum[n] *= exp(local max - global max)
__global__ __launch_bounds__(256) void k_shard_rescale(int N, const float* __restrict__ lmax, const float* __restrict__ gmax,
                                                       float* __restrict__ psum) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) psum[n] *= expf(lmax[n] - gmax[n]);
}

// shard mode: per-row loss and rank from the GLOBAL max / sum-exp / rank (identical on every shard)
template <int KIND>
__global__ __launch_bounds__(256) void k_shard_finish(HeadConst h, int N, const float* __restrict__ xnorm,
                                                      const float* __restrict__ ty, const float* __restrict__ state_t,
                                                      const float* __restrict__ rowp, const float* __restrict__ gmax,
                                                      const float* __restrict__ gsum, const float* __restrict__ grank,
                                                      float* __restrict__ lse_ws, float* __restrict__ rowloss,
                                                      int32_t* __restrict__ rowrank) {
  if (KIND == FRX_SPHERE && (h.flags & 4)) h.lamb = *state_t;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const RowCtx r = make_row_ctx(KIND, n, h, xnorm, ty, state_t, rowp);
  float zy, d, u;
  head_z<KIND>(ty[n], true, h, r, zy, d, u);
  const float lse = gmax[n] + logf(gsum[n]);
  lse_ws[n] = lse;
  rowloss[n] = lse - zy;
  rowrank[n] = (int)(grank[n] + 0.5f);
}

// loss = mean(rowloss); topk = (#rank<1, #rank<5).  Single block: deterministic order.
__global__ __launch_bounds__(256) void k_head_finalize(const float* __restrict__ rowloss,
                                                       const int32_t* __restrict__ rowrank, int N,
                                                       float* __restrict__ loss, int32_t* __restrict__ topk,
                                                       const float* __restrict__ lse_ws, float* __restrict__ lse_out,
                                                       const float* __restrict__ norms_ws, float* __restrict__ norms_out) {
  __shared__ float sh[4];
  __shared__ int s1[4], s5[4];
  float a = 0.f;
  int c1 = 0, c5 = 0;
  for (int n = threadIdx.x; n < N; n += 256) {
    a += rowloss[n];
    c1 += rowrank[n] < 1;
    c5 += rowrank[n] < 5;
    // the caller's copies of the per-row outputs ride along (each was a dependent launch of its own for N floats)
    if (lse_out) lse_out[n] = lse_ws[n];
    if (norms_out) norms_out[n] = norms_ws[n];
  }
  a = block_sum256(a, sh);
  c1 = wave_sum_i(c1);
  c5 = wave_sum_i(c5);
  if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = c1; s5[threadIdx.x >> 6] = c5; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (loss) *loss = a / (float)N;
    if (topk) { topk[0] = s1[0] + s1[1] + s1[2] + s1[3]; topk[1] = s5[0] + s5[1] + s5[2] + s5[3]; }
  }
}

// Backward row sweep: dC[n][j] = (softmax_j - 1[j==y]) * gscale * dz/dc * clamp-pass.
template <int KIND>
__global__ __launch_bounds__(256) void k_head_grad(HeadConst h, const float* __restrict__ cbuf, int C,
                                                   int Cpad, const int64_t* __restrict__ labels,
                                                   const float* __restrict__ xnorm,
                                                   const float* __restrict__ ty,
                                                   const float* __restrict__ state_t,
                                                   const float* __restrict__ rowp,
                                                   const float* __restrict__ lse,
                                                   const float* __restrict__ gout, float inv_n,
                                                   const float* __restrict__ dlogits,
                                                   float* __restrict__ gbuf, float* __restrict__ dn,
                                                   float* __restrict__ zero_rows, int zero_len) {
  __shared__ float sh[4];
  if (KIND == FRX_SPHERE && (h.flags & 4)) h.lamb = *state_t;
  const int n = blockIdx.x;
  // row n of the split-K accumulator of the dX GEMM that follows is cleared here (it was a launch of its own; a kernel,
  // not a memset node: see head_bwd_impl)
  for (int j = threadIdx.x; j < zero_len; j += 256) zero_rows[(long)n * zero_len + j] = 0.f;
  bool bad_label, owned;
  const int y = shard_label(labels[n], h, C, owned, bad_label);
  const float* crow = cbuf + (long)n * Cpad;
  float* grow = gbuf + (long)n * Cpad;
  const RowCtx r = make_row_ctx(KIND, n, h, xnorm, ty, state_t, rowp);
  const float gs = (gout ? *gout : 1.f) * inv_n;
  const float l = lse[n];
  float dnorm = 0.f;
  constexpr int HU = 4;                                            // 16-byte loads / stores, HU trips in flight, as in k_head_rows
  for (int j0 = threadIdx.x * 4; j0 < Cpad; j0 += 1024 * HU) {
    float4 c4[HU];
#pragma unroll
    for (int t = 0; t < HU; ++t) c4[t] = *reinterpret_cast<const float4*>(crow + (j0 + 1024 * t < Cpad ? j0 + 1024 * t : j0));
#pragma unroll
    for (int t = 0; t < HU; ++t) {
      const int jt = j0 + 1024 * t;
      if (jt >= Cpad) break;
      const float cv[4] = {c4[t].x, c4[t].y, c4[t].z, c4[t].w};
      float ov[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = jt + e;
        float out = 0.f;
        if (j < C) {
          const float craw = cv[e];
          const float cc = head_clamp<KIND>(craw);
          float z, d, u;
          head_z<KIND>(cc, j == y, h, r, z, d, u);
          // upstream gradient: mean-CE in closed form, or an arbitrary dL/dlogits supplied by autograd
          const float g = dlogits ? dlogits[(long)n * C + j] : (expf(z - l) - (j == y ? 1.f : 0.f)) * gs;
          out = head_pass<KIND>(craw) ? g * d : 0.f;
          if (KIND == FRX_SPHERE) dnorm += g * u;
          if (KIND == FRX_MAG && j == y) dnorm += g * u * r.aux;      // through the adaptive margin (criterion.py:1264)
        }
        ov[e] = out;
      }
      *reinterpret_cast<float4*>(grow + jt) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
  }
  if (KIND == FRX_SPHERE || KIND == FRX_MAG) {
    dnorm = block_sum256(dnorm, sh);
    if (threadIdx.x == 0) {
      if (KIND == FRX_MAG && r.xnorm >= h.p2 && r.xnorm <= h.p3) {   // + lamb * d loss_g / d||x||  (criterion.py:1235-1239); gs = 1/N in dlogits mode
        const float xn = r.xnorm;                               // clamp inactive here, so x_norm == ||x||
        dnorm += h.lamb * gs * (1.f / (h.p3 * h.p3) - 1.f / (xn * xn));
      }
      dn[n] = dnorm;
    }
  }
}

// d(normalise) for row vectors: out = (dh - h*(h.dh)) * inv [+ dn * h], h = a*inv.  Wave per row.
__global__ __launch_bounds__(256) void k_norm_bwd_rows(const float* __restrict__ a,
                                                       const float* __restrict__ dh,
                                                       const float* __restrict__ inv,
                                                       const float* __restrict__ dn, int R, int D,
                                                       float* __restrict__ out, int accumulate) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float iv = inv[row];
  const float* pa = a + (long)row * D;
  const float* pd = dh + (long)row * D;
  float dot = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 x = *reinterpret_cast<const float4*>(pa + d);
    const float4 g = *reinterpret_cast<const float4*>(pd + d);
    dot += (x.x * g.x + x.y * g.y + x.z * g.z + x.w * g.w) * iv;
  }
  dot = wave_sum(dot);
  const float extra = dn ? dn[row] : 0.f;
  float* po = out + (long)row * D;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 x = *reinterpret_cast<const float4*>(pa + d);
    const float4 g = *reinterpret_cast<const float4*>(pd + d);
    float4 o;
    o.x = (g.x - x.x * iv * dot) * iv + extra * x.x * iv;
    o.y = (g.y - x.y * iv * dot) * iv + extra * x.y * iv;
    o.z = (g.z - x.z * iv * dot) * iv + extra * x.z * iv;
    o.w = (g.w - x.w * iv * dot) * iv + extra * x.w * iv;
    if (accumulate) {
      const float4 p = *reinterpret_cast<const float4*>(po + d);
      o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
    }
    *reinterpret_cast<float4*>(po + d) = o;
  }
}

// same for column vectors of a [D, C] matrix: 64 columns x 4 row-lanes per block (see k_col_norms)
__global__ __launch_bounds__(256) void k_norm_bwd_cols(const float* __restrict__ a,
                                                       const float* __restrict__ dh,
                                                       const float* __restrict__ inv, int D, int C,
                                                       float* __restrict__ out, int accumulate) {
  __shared__ float red[COLR][COLB];
  const int cl = threadIdx.x % COLB, rl = threadIdx.x / COLB;
  const int c = blockIdx.x * COLB + cl;
  const bool live = c < C;
  const float iv = live ? inv[c] : 0.f;
  float part = 0.f;
  if (live) {
    for (int d0 = rl; d0 < D; d0 += COLR * COLU) {
      float va[COLU], vd[COLU];
#pragma unroll
      for (int u = 0; u < COLU; ++u) {
        const int d = d0 + COLR * u;
        const long i = (long)(d < D ? d : d0) * C + c;
        va[u] = a[i]; vd[u] = d < D ? dh[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < COLU; ++u) part += va[u] * iv * vd[u];
    }
  }
  red[rl][cl] = part;
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < COLR; ++i) dot += red[i][cl];
  if (!live) return;
  for (int d0 = rl; d0 < D; d0 += COLR * COLU) {
    float va[COLU], vd[COLU], vo[COLU];
#pragma unroll
    for (int u = 0; u < COLU; ++u) {
      const int d = d0 + COLR * u;
      const long i = (long)(d < D ? d : d0) * C + c;
      va[u] = a[i]; vd[u] = dh[i]; vo[u] = accumulate ? out[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < COLU; ++u) {
      const int d = d0 + COLR * u;
      if (d < D) out[(long)d * C + c] = (vd[u] - va[u] * iv * dot) * iv + vo[u];
    }
  }
}

// The same in ONE pass over memory for D <= NB_ROWS * NB_RL (the 512-d embedding): 64 columns x 16 row-lanes per
// 1024-thread block, a thread's <= 32 rows of both operands stay in registers between the dot product and the output.
// The two-pass form re-read 2 x 174 MB at C = 85 000 from HBM (a block's 256 KB does not survive in L2): 177 -> ~105 us.
constexpr int NB_RL = 16, NB_ROWS = 32;
__global__ __launch_bounds__(1024) void k_norm_bwd_cols_reg(const float* __restrict__ a, const float* __restrict__ dh,
                                                            const float* __restrict__ inv, int D, int C,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ float red[NB_RL][COLB];
  const int cl = threadIdx.x % COLB, rl = threadIdx.x / COLB;
  const int c = blockIdx.x * COLB + cl;
  const bool live = c < C;
  const float iv = live ? inv[c] : 0.f;
  float va[NB_ROWS], vd[NB_ROWS];
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < NB_ROWS; ++u) {
    const int d = rl + NB_RL * u;
    const bool ok = live && d < D;
    const long i = ok ? (long)d * C + c : 0;
    va[u] = ok ? a[i] : 0.f;
    vd[u] = ok ? dh[i] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NB_ROWS; ++u) part += va[u] * iv * vd[u];
  red[rl][cl] = part;
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NB_RL; ++i) dot += red[i][cl];
  if (!live) return;
#pragma unroll
  for (int u = 0; u < NB_ROWS; ++u) {
    const int d = rl + NB_RL * u;
    if (d < D) {
      const long i = (long)d * C + c;
      out[i] = (vd[u] - va[u] * iv * dot) * iv + (accumulate ? out[i] : 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void k_copy_f32(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = x[i];
}

// ------------------------------------------------------------------------------------------
// Round 4 (north_star: "MFMA GEMM + fused row-max/exp/sum epilogue"): the cosine GEMM's epilogue evaluates the margin and the
// softmax statistics of its own 64 x 64 tile -- per row: max logit, sum of exp(logit - max), number of pre-margin cosines
// above the row's target -- and one combine launch closes loss / lse / top-k from the per-tile partials (the arithmetic of
// the class-sharded head's phases, with column tiles in the role of ranks): the train path's forward no longer sweeps the
// [N, C] cosines (k_head_rows) and no longer runs k_head_ty / k_head_finalize.  The cosines are still WRITTEN: the backward
// reads them (recomputing them inside the dX and dW products instead would cost two more cosine GEMMs).
// The row context needs the target cosine BEFORE the GEMM: k_head_ty_dot evaluates it as N dot products of 512 and the
// GEMM's epilogue writes that very value into element (n, y_n), so the matrix and the per-row vector stay one number.
// Kinds whose row context is complete in the cosine phase: ARC, COS, MV_AM, MV_ARC (SphereFace's annealing lambda, CurricularFace's
// EMA, AdaFace's / MagFace's batch statistics and the elastic margins' rank matching arrive in the loss phase).
// ------------------------------------------------------------------------------------------
// ty <- tyg (may be the same array), ty_sum <- their sum in a fixed order
__global__ __launch_bounds__(256) void k_shard_take_ty(const float* tyg, int N, float* ty, float* ty_sum) {
  __shared__ float sh[4];
  float part = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) { const float v = tyg[n]; ty[n] = v; part += v; }
  const float tot = block_sum256(part, sh);
  if (threadIdx.x == 0) *ty_sum = tot;
}

static bool head_epi_kind(int kind) {
  return kind == FRX_ARC || kind == FRX_COS || kind == FRX_MV_AM || kind == FRX_MV_ARC;
}
// (not in class-sharded mode: a row's target may live on another rank)
static bool head_fused(const frx_head_desc* d) { return head_epi_kind(d->kind) && !(d->flags & 8); }

// raw[n] = ((x_n . w_y) * winv[y]) * xinv[n] (the GEMM epilogue's order of scalings); ty[n] = clamped.  One wave per row.
template <int KIND>
__global__ __launch_bounds__(256) void k_head_ty_dot(const float* __restrict__ x, const float* __restrict__ w, int w_cd,
                                                     const int64_t* __restrict__ labels, const float* __restrict__ xinv,
                                                     const float* __restrict__ winv, int N, int D, int C, float* __restrict__ ty,
                                                     float* __restrict__ ty_raw) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  bool bad;
  const int y = safe_label(labels[n], C, bad);
  float acc = 0.f;
  for (int dd = lane; dd < D; dd += 64) acc = fmaf(x[(long)n * D + dd], w_cd ? w[(long)y * D + dd] : w[(long)dd * C + y], acc);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  const float raw = (acc * winv[y]) * xinv[n];
  if (lane == 0) { ty[n] = bad ? NAN : head_clamp<KIND>(raw); ty_raw[n] = raw; }
}

struct HeadEpi {          // what the cosine GEMM's fused epilogue needs besides GemmArgs
  HeadConst h;
  const int64_t* labels;
  const float* xnorm;     // [N]
  const float* ty;        // [N] clamped target cosines (k_head_ty_dot)
  const float* ty_raw;    // [N] the unclamped values: written into element (n, y_n) of the matrix
  float* part;            // [tilesN][3][Npad]: max | sum-exp | rank of each row over the tile's columns
  int Npad;
};

template <bool BNC, int KIND>
__global__ __launch_bounds__(256) void k_gemm_f32_head(GemmArgs g, HeadEpi e) {
  // (one array: the epilogue re-uses the operand stages as a 64 x 65 float image of the finished tile)
  __shared__ __attribute__((aligned(16))) float S[2 * 2 * GBK * GLD];
  float (*As)[GBK][GLD] = reinterpret_cast<float (*)[GBK][GLD]>(S);
  float (*Bs)[GBK][GLD] = reinterpret_cast<float (*)[GBK][GLD]>(S + 2 * GBK * GLD);
  static_assert(2 * 2 * GBK * GLD >= GBM * (GBN + 1), "the tile image must fit the operand stages");
  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
  gemm_f32_tile<false, BNC>(g, m0, n0, 0, g.K, As, Bs, acc);      // (ends on a barrier: the stages are free)
  // phase A, MFMA layout (lane = column, 16 rows per lane): scale, put the dot product's own value into the target element,
  // store the cosine, and lay the tile out in LDS
  {
    const int nl = wn * 32 + li, n = n0 + nl;
    const bool ncol = n < g.N;
    const float cs = ncol ? g.col_scale[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = m0 + ml;
      const bool mrow = m < g.M;
      const int ms = mrow ? m : 0;
      float v = acc[r] * cs * g.row_scale[ms];
      if (ncol && (int64_t)n == e.labels[ms]) v = e.ty_raw[ms];      // one number for the matrix and the per-row vector
      if (mrow && ncol) g.C[(long)m * g.ldc + n] = v;
      S[ml * (GBN + 1) + nl] = v;
    }
  }
  __syncthreads();
  // phase B, row layout: four threads per row, sixteen columns each -- margin, running max, sum of exp, rank count
  {
    const int ml = threadIdx.x >> 2, q = threadIdx.x & 3, m = m0 + ml;
    const int ms = m < g.M ? m : 0;
    bool bad;
    const int y = safe_label(e.labels[ms], g.N, bad);
    RowCtx rc;
    rc.xnorm = e.xnorm[ms]; rc.t = 0.f; rc.p = 0.f; rc.aux = 0.f; rc.ty = e.ty[ms];
    rc.cm = KIND == FRX_MV_AM ? rc.ty - e.h.m : (KIND == FRX_MV_ARC ? rc.ty * e.h.cos_m - sqrtf(1.f - rc.ty * rc.ty + 1e-9f) * e.h.sin_m : 0.f);
    const float cs_y = head_cos_s<KIND>(rc.ty, e.h, rc);
    float z[16];
    float zmax = -INFINITY, rk = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int nl = q * 16 + c, n = n0 + nl;
      const float cc = head_clamp<KIND>(S[ml * (GBN + 1) + nl]);
      float dd, u;
      head_z<KIND>(cc, n == y && !bad, e.h, rc, z[c], dd, u);
      if (n >= g.N) z[c] = -INFINITY;
      else rk += head_cos_s<KIND>(cc, e.h, rc) > cs_y ? 1.f : 0.f;
      zmax = fmaxf(zmax, z[c]);
    }
    zmax = fmaxf(zmax, __shfl_xor(zmax, 1, 64));
    zmax = fmaxf(zmax, __shfl_xor(zmax, 2, 64));
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) se += expf(z[c] - zmax);      // (columns past C: exp(-inf) = 0; a tile always holds a real column)
    se += __shfl_xor(se, 1, 64); rk += __shfl_xor(rk, 1, 64);
    se += __shfl_xor(se, 2, 64); rk += __shfl_xor(rk, 2, 64);
    if (q == 0 && m < g.M) {
      float* p = e.part + (long)blockIdx.x * 3 * e.Npad;
      p[m] = zmax; p[e.Npad + m] = se; p[2 * e.Npad + m] = rk;
    }
  }
}

// closes the fused forward: one WAVE per row reduces the partials of every column tile (lanes stride over the tiles) -> lse,
// row loss, rank; k_head_finalize then takes the batch mean / top-k counts in its fixed order
template <int KIND>
__global__ __launch_bounds__(256) void k_head_combine(HeadConst h, int N, int Npad, int tiles, const float* __restrict__ part,
                                                      const int64_t* __restrict__ labels, int C, const float* __restrict__ xnorm,
                                                      const float* __restrict__ ty, float* __restrict__ lse_ws,
                                                      float* __restrict__ rowloss, int32_t* __restrict__ rowrank) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float mx = -INFINITY;
  for (int t = lane; t < tiles; t += 64) mx = fmaxf(mx, part[(long)t * 3 * Npad + n]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f, rk = 0.f;
  for (int t = lane; t < tiles; t += 64) {
    const float* p = part + (long)t * 3 * Npad;
    sum += p[Npad + n] * expf(p[n] - mx);
    rk += p[2 * Npad + n];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { sum += __shfl_xor(sum, o, 64); rk += __shfl_xor(rk, o, 64); }
  if (lane == 0) {
    RowCtx rc;
    rc.xnorm = xnorm[n]; rc.t = 0.f; rc.p = 0.f; rc.aux = 0.f; rc.ty = ty[n];
    rc.cm = KIND == FRX_MV_AM ? rc.ty - h.m : (KIND == FRX_MV_ARC ? rc.ty * h.cos_m - sqrtf(1.f - rc.ty * rc.ty + 1e-9f) * h.sin_m : 0.f);
    float zy, d, u;
    head_z<KIND>(ty[n], true, h, rc, zy, d, u);
    bool bad;
    safe_label(labels[n], C, bad);
    const float lse = mx + logf(sum);
    lse_ws[n] = lse;
    rowloss[n] = bad ? NAN : lse - zy;
    rowrank[n] = (int)(rk + 0.5f);
  }
}

// ------------------------------------------------------------------------------------------
// Workspace carving
// ------------------------------------------------------------------------------------------
struct HeadWs {
  float *xinv, *xnorm, *winv, *ty, *tysum, *rowloss, *lse, *dn, *rowp, *xn, *lossg, *cbuf, *gbuf, *dxh, *dwh;
  float *minv, *cbuf2, *gbuf2;     // VPL: inverse norms of the memory rows, cosine against the memory, its gradient
  float *tyraw, *part;             // fused forward: unclamped target cosines [Npad]; per-tile softmax partials [tilesN][3][Npad]
  int32_t* rowrank;
  int Cpad, Npad;
  size_t bytes;
};

static HeadWs carve(const frx_head_desc* d, void* base) {
  HeadWs w;
  w.Cpad = (int)round_up(d->C, 64);
  w.Npad = (int)round_up(d->N, 64);
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* r = (float*)(p + off);
    off += round_up(nfloat * sizeof(float), 256);
    return r;
  };
  w.xinv = take(w.Npad); w.xnorm = take(w.Npad); w.winv = take(w.Cpad); w.ty = take(w.Npad);
  w.tysum = take(64); w.rowloss = take(w.Npad); w.lse = take(w.Npad); w.dn = take(w.Npad);
  w.rowrank = (int32_t*)take(w.Npad);
  w.rowp = take(w.Npad); w.xn = take(w.Npad); w.lossg = take(64);
  w.cbuf = take((size_t)d->N * w.Cpad);
  w.gbuf = take((size_t)d->N * w.Cpad);
  w.dxh = take((size_t)d->N * d->D);
  w.dwh = take((size_t)d->C * d->D);
  w.tyraw = w.part = nullptr;
  if (head_epi_kind(d->kind) && !(d->flags & 8)) {      // (head_fused())
    w.tyraw = take(w.Npad);
    w.part = take((size_t)cdiv(d->C, GBN) * 3 * w.Npad);
  }
  w.minv = w.cbuf2 = w.gbuf2 = nullptr;
  if (d->kind == FRX_VPL) {
    w.minv = take(w.Cpad);
    w.cbuf2 = take((size_t)d->N * w.Cpad);
    w.gbuf2 = take((size_t)d->N * w.Cpad);
  }
  w.bytes = off;
  return w;
}

static int check_desc(const frx_head_desc* d) {
  FRX_CHECK_ARG(d != nullptr, "head desc is NULL");
  FRX_CHECK_ARG(d->kind >= FRX_ARC && d->kind <= FRX_VPL, "unknown head kind %d", d->kind);
  FRX_CHECK_ARG(d->N > 0 && d->C > 0 && d->D > 0, "head dims must be positive (N=%d D=%d C=%d)", d->N, d->D, d->C);
  FRX_CHECK_ARG(d->D % 16 == 0, "head feature dim D=%d must be a multiple of 16", d->D);
  FRX_CHECK_ARG(d->kind != FRX_SPHERE || (d->m == 1.f || d->m == 2.f || d->m == 3.f || d->m == 4.f || d->m == 5.f),
                "SphereFace's margin is an integer 1..5 (the Chebyshev table of criterion.py:40-47), got %g", (double)d->m);
  FRX_CHECK_ARG(d->kind != FRX_MAG || (d->p[3] > d->p[2] && d->p[2] > 0.f), "MagFace needs 0 < l_a < u_a (got %g, %g)",
                (double)d->p[2], (double)d->p[3]);
  FRX_CHECK_ARG(d->kind != FRX_ADA || d->N > 1, "AdaFace's batch std needs N > 1");
  if (d->flags & 8) {
    FRX_CHECK_ARG(d->kind == FRX_ARC || d->kind == FRX_COS || d->kind == FRX_SPHERE || d->kind == FRX_CURR || d->kind == FRX_MV_AM ||
                  d->kind == FRX_MV_ARC, "class-sharded mode supports ARC / COS / SPHERE / CURR / MV_* (kind %d keeps batch- or class-wide state)", d->kind);
    FRX_CHECK_ARG(d->class_offset >= 0, "class_offset must be >= 0");
  }
  return FRX_OK;
}

static HeadConst make_const(const frx_head_desc* d) {
  HeadConst h;
  h.kind = d->kind;
  h.s = d->s;
  h.m = d->m;
  h.cos_m = (float)cos((double)d->m);
  h.sin_m = (float)sin((double)d->m);
  h.th = (float)cos(M_PI - (double)d->m);
  h.mm = (float)(sin(M_PI - (double)d->m) * (double)d->m);
  h.lamb = d->lamb;
  h.p0 = d->p[0]; h.p1 = d->p[1]; h.p2 = d->p[2]; h.p3 = d->p[3];
  h.flags = d->flags;
  h.c0 = (d->flags & 8) ? d->class_offset : 0;
  return h;
}

static bool w_is_cd(int kind) { return kind == FRX_ARC || kind == FRX_SPHERE || kind == FRX_MV_AM || kind == FRX_MV_ARC || kind == FRX_VPL; }
static bool vpl_memory_on(const frx_head_desc* d) { return d->kind == FRX_VPL && (d->flags & 2); }
static bool needs_state(int kind) {
  return kind == FRX_CURR || kind == FRX_ADA || kind == FRX_ELASTIC_ARC || kind == FRX_ELASTIC_COS || kind == FRX_VPL;
}
static bool needs_state_d(const frx_head_desc* d) { return needs_state(d->kind) || (d->kind == FRX_SPHERE && (d->flags & 4)); }
static const char* state_what(int kind) {
  return kind == FRX_CURR ? "CurricularFace needs the `t` buffer"
       : kind == FRX_ADA  ? "AdaFace needs its [batch_mean, batch_std] state"
       : kind == FRX_VPL  ? "VPL-ArcFace needs its [mem | life] state"
       : kind == FRX_SPHERE ? "SphereFace with flags bit 2 reads its lambda from state_t[0]"
                          : "the elastic heads need this step's per-row margins";
}
// expands M(KIND) for the runtime kind
#define FRX_KIND_SWITCH(kind, M)                          \
  switch (kind) {                                         \
    case FRX_ARC: M(FRX_ARC); break;                      \
    case FRX_COS: M(FRX_COS); break;                      \
    case FRX_SPHERE: M(FRX_SPHERE); break;                \
    case FRX_CURR: M(FRX_CURR); break;                    \
    case FRX_MV_AM: M(FRX_MV_AM); break;                  \
    case FRX_MV_ARC: M(FRX_MV_ARC); break;                \
    case FRX_ADA: M(FRX_ADA); break;                      \
    case FRX_ELASTIC_ARC: M(FRX_ELASTIC_ARC); break;      \
    case FRX_ELASTIC_COS: M(FRX_ELASTIC_COS); break;      \
    case FRX_VPL: M(FRX_VPL); break;                      \
    default: M(FRX_MAG); break;                           \
  }

}  // namespace frx

using namespace frx;

extern "C" size_t frx_head_workspace_bytes(const frx_head_desc* d) {
  if (check_desc(d) != FRX_OK) return 0;
  return carve(d, nullptr).bytes;
}

extern "C" int frx_head_fwd_cos(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                                const float* w, const int64_t* labels, void* ws, size_t ws_bytes,
                                float* ty_sum_out) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(x && w && labels && ws, "head_fwd_cos: NULL pointer");
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  hipLaunchKernelGGL(k_row_norms, dim3(cdiv(d->N, 4)), dim3(256), 0, st, x, d->N, d->D, W.xinv, W.xnorm);
  FRX_LAUNCH_CHECK();
  if (w_is_cd(d->kind))
    hipLaunchKernelGGL(k_row_norms, dim3(cdiv(d->C, 4)), dim3(256), 0, st, w, d->C, d->D, W.winv, (float*)nullptr);
  else
    hipLaunchKernelGGL(k_col_norms, dim3(cdiv(d->C, COLB)), dim3(256), 0, st, w, d->D, d->C, W.winv);
  FRX_LAUNCH_CHECK();
  GemmArgs g{};
  g.A = x; g.lda = d->D; g.a_mcontig = 0;
  g.B = w; g.M = d->N; g.N = d->C; g.K = d->D;
  if (w_is_cd(d->kind)) { g.b_ncontig = 0; g.ldb = d->D; } else { g.b_ncontig = 1; g.ldb = d->C; }
  g.C = W.cbuf; g.ldc = W.Cpad; g.row_scale = W.xinv; g.col_scale = W.winv;
  float* tys = ty_sum_out ? ty_sum_out : W.tysum;
  const HeadConst hc = make_const(d);
  if (head_fused(d)) {
    // the fused forward (see k_gemm_f32_head): target cosines first, then the GEMM whose epilogue leaves the per-tile softmax
    // partials frx_head_fwd_loss combines (when the caller asks for no [N, C] outputs) -- or ignores (k_head_rows then)
    const int cd = w_is_cd(d->kind) ? 1 : 0;
#define FRX_TYD(K) hipLaunchKernelGGL(k_head_ty_dot<K>, dim3(cdiv(d->N, 4)), dim3(256), 0, st, x, w, cd, labels, (const float*)W.xinv, \
                                      (const float*)W.winv, d->N, d->D, d->C, W.ty, W.tyraw)
    FRX_KIND_SWITCH(d->kind, FRX_TYD)
#undef FRX_TYD
    if (ty_sum_out)       // (only CurricularFace consumes the sum, and it does not take this path: on request only)
      hipLaunchKernelGGL(k_shard_take_ty, dim3(1), dim3(256), 0, st, (const float*)W.ty, d->N, W.ty, tys);
    HeadEpi e{hc, labels, W.xnorm, W.ty, W.tyraw, W.part, W.Npad};
    g.ksplit_len = (int)round_up((size_t)g.K, GBK);
    dim3 grid(cdiv(g.N, GBN), cdiv(g.M, GBM), 1), block(256);
#define FRX_GH(K) do { if (cd) hipLaunchKernelGGL((k_gemm_f32_head<false, K>), grid, block, 0, st, g, e); \
                       else hipLaunchKernelGGL((k_gemm_f32_head<true, K>), grid, block, 0, st, g, e); } while (0)
    switch (d->kind) {
      case FRX_ARC: FRX_GH(FRX_ARC); break;
      case FRX_COS: FRX_GH(FRX_COS); break;
      case FRX_MV_AM: FRX_GH(FRX_MV_AM); break;
      default: FRX_GH(FRX_MV_ARC); break;
    }
#undef FRX_GH
    FRX_LAUNCH_CHECK();
    return FRX_OK;
  }
  if (int rc = launch_gemm(st, g, 1)) return rc;
#define FRX_TY(K) hipLaunchKernelGGL(k_head_ty<K>, dim3(1), dim3(256), 0, st, hc, W.cbuf, d->N, (long)W.Cpad, d->C, labels, W.ty, tys)
  FRX_KIND_SWITCH(d->kind, FRX_TY)
#undef FRX_TY
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_head_vpl_prepare(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                                    const int64_t* labels, float* state_t, void* ws, size_t ws_bytes) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(d->kind == FRX_VPL, "head_vpl_prepare: kind %d is not FRX_VPL", d->kind);
  if (!vpl_memory_on(d)) return FRX_OK;
  FRX_CHECK_ARG(x && labels && state_t && ws, "head_vpl_prepare: NULL pointer");
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  float* mem = state_t;
  float* life = state_t + (size_t)d->C * d->D;
  hipLaunchKernelGGL(k_vpl_mem_update, dim3(d->N), dim3(256), 0, st, x, labels, d->N, d->D, d->C, d->p[1], mem, life);
  hipLaunchKernelGGL(k_vpl_life_decay, dim3(cdiv(d->C, 256)), dim3(256), 0, st, life, d->C);
  hipLaunchKernelGGL(k_row_norms, dim3(cdiv(d->C, 4)), dim3(256), 0, st, (const float*)mem, d->C, d->D, W.minv, (float*)nullptr);
  FRX_LAUNCH_CHECK();
  GemmArgs g{};
  g.A = x; g.lda = d->D; g.a_mcontig = 0;
  g.B = mem; g.M = d->N; g.N = d->C; g.K = d->D; g.b_ncontig = 0; g.ldb = d->D;
  g.C = W.cbuf2; g.ldc = W.Cpad; g.row_scale = W.xinv; g.col_scale = W.minv;
  if (int rc = launch_gemm(st, g, 1)) return rc;
  hipLaunchKernelGGL(k_vpl_blend, dim3(d->N), dim3(256), 0, st, W.cbuf, (const float*)W.cbuf2, (const float*)life, labels,
                     d->C, (long)W.Cpad, d->p[0]);
  hipLaunchKernelGGL(k_head_ty<FRX_VPL>, dim3(1), dim3(256), 0, st, make_const(d), W.cbuf, d->N, (long)W.Cpad, d->C, labels, W.ty, W.tysum);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_head_fwd_loss(int device, frx_stream_t stream, const frx_head_desc* d,
                                 const int64_t* labels, float* state_t, const float* ty_sum,
                                 int64_t ty_count, void* ws, size_t ws_bytes, float* cos_s, float* logits,
                                 float* norms, float* loss, float* lse, int32_t* topk) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(labels && ws, "head_fwd_loss: NULL pointer");
  FRX_CHECK_ARG(!needs_state_d(d) || state_t, "%s", state_what(d->kind));
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  HeadConst h = make_const(d);
  if (d->kind == FRX_CURR) {
    const float* src = ty_sum ? ty_sum : W.tysum;
    const double cnt = ty_sum ? (double)ty_count : (double)d->N;
    hipLaunchKernelGGL(k_curr_t_update, dim3(1), dim3(64), 0, st, state_t, src, (float)(1.0 / cnt), d->momentum);
    FRX_LAUNCH_CHECK();
  }
  if (d->kind == FRX_ADA)
    hipLaunchKernelGGL(k_head_rowparam<FRX_ADA>, dim3(1), dim3(256), 0, st, h, (const float*)W.xnorm, d->N, state_t, W.rowp, W.xn, W.lossg);
  if (d->kind == FRX_MAG)
    hipLaunchKernelGGL(k_head_rowparam<FRX_MAG>, dim3(1), dim3(256), 0, st, h, (const float*)W.xnorm, d->N, state_t, W.rowp, W.xn, W.lossg);
  FRX_LAUNCH_CHECK();
  FRX_CHECK_ARG(!(d->flags & 8), "head_fwd_loss: a class-sharded head (flags bit 3) runs frx_head_shard_rows / _finish instead");
  const float* rowp = (d->kind == FRX_ELASTIC_ARC || d->kind == FRX_ELASTIC_COS) ? (const float*)state_t : (const float*)W.rowp;
  if (head_fused(d) && !cos_s && !logits) {      // the cosine GEMM's epilogue left per-tile partials: one combine launch closes
#define FRX_COMB(K)                                                                                                      \
    hipLaunchKernelGGL(k_head_combine<K>, dim3(cdiv(d->N, 4)), dim3(256), 0, st, h, d->N, W.Npad, cdiv(d->C, GBN), (const float*)W.part, \
                       labels, d->C, (const float*)W.xnorm, (const float*)W.ty, W.lse, W.rowloss, W.rowrank)
    switch (d->kind) {
      case FRX_ARC: FRX_COMB(FRX_ARC); break;
      case FRX_COS: FRX_COMB(FRX_COS); break;
      case FRX_MV_AM: FRX_COMB(FRX_MV_AM); break;
      default: FRX_COMB(FRX_MV_ARC); break;
    }
#undef FRX_COMB
    hipLaunchKernelGGL(k_head_finalize, dim3(1), dim3(256), 0, st, (const float*)W.rowloss, (const int32_t*)W.rowrank, d->N, loss, topk,
                       (const float*)W.lse, lse, (const float*)W.xnorm, norms);
    FRX_LAUNCH_CHECK();
    return FRX_OK;
  }
#define FRX_ROWS(K)                                                                                   \
  hipLaunchKernelGGL(k_head_rows<K>, dim3(d->N), dim3(256), 0, st, h, (const float*)W.cbuf, d->C,     \
                     (long)W.Cpad, labels, (const float*)W.xnorm, (const float*)W.ty,                 \
                     (const float*)state_t, rowp, cos_s, logits, W.lse, W.rowloss, W.rowrank, (float*)nullptr)
  FRX_KIND_SWITCH(d->kind, FRX_ROWS)
#undef FRX_ROWS
  FRX_LAUNCH_CHECK();
  // (lse / norms are copied by the same kernel: no memcpy nodes in a captured step -- see head_bwd_impl -- and no launches
  // of their own)
  hipLaunchKernelGGL(k_head_finalize, dim3(1), dim3(256), 0, st, (const float*)W.rowloss,
                     (const int32_t*)W.rowrank, d->N, loss, topk, (const float*)W.lse, lse,
                     (const float*)(d->kind == FRX_MAG ? W.xn : W.xnorm), norms);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

// After frx_head_fwd_cos: the clamped target cosine of every row (what the elastic heads' plus=True variant ranks its
// margins by, criterion.py:1006-1011 / 1117-1122).
extern "C" int frx_head_target_cos(int device, frx_stream_t stream, const frx_head_desc* d, void* ws, size_t ws_bytes,
                                   float* ty_out) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(ws && ty_out, "head_target_cos: NULL pointer");
  FRX_ENTER(device);
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  hipLaunchKernelGGL(k_copy_f32, dim3(cdiv(d->N, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)W.ty, ty_out, (long)d->N);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

// ---------------------------------------------------------------------------------------- class-sharded head
extern "C" int frx_head_shard_cos(int device, frx_stream_t stream, const frx_head_desc* d, const float* x, const float* w,
                                  const int64_t* labels, void* ws, size_t ws_bytes, float* ty_out) {
  FRX_CHECK_ARG(d && (d->flags & 8), "head_shard_cos: the descriptor is not in class-sharded mode (flags bit 3)");
  FRX_CHECK_ARG(ty_out != nullptr, "head_shard_cos: ty_out is NULL");
  if (int rc = frx_head_fwd_cos(device, stream, d, x, w, labels, ws, ws_bytes, nullptr)) return rc;
  FRX_ENTER(device);
  HeadWs W = carve(d, ws);
  hipLaunchKernelGGL(k_copy_f32, dim3(cdiv(d->N, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)W.ty, ty_out, (long)d->N);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}


extern "C" int frx_head_shard_rows(int device, frx_stream_t stream, const frx_head_desc* d, const int64_t* labels,
                                   float* state_t, const float* ty_global, void* ws, size_t ws_bytes, float* part) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(d->flags & 8, "head_shard_rows: the descriptor is not in class-sharded mode (flags bit 3)");
  FRX_CHECK_ARG(labels && ty_global && ws && part, "head_shard_rows: NULL pointer");
  FRX_CHECK_ARG(!needs_state_d(d) || state_t, "%s", state_what(d->kind));
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  HeadConst h = make_const(d);
  // the all-reduced target cosines replace this shard's partial vector; their sum over ALL rows feeds CurricularFace's EMA
  hipLaunchKernelGGL(k_shard_take_ty, dim3(1), dim3(256), 0, st, ty_global, d->N, W.ty, W.tysum);
  if (d->kind == FRX_CURR)
    hipLaunchKernelGGL(k_curr_t_update, dim3(1), dim3(64), 0, st, state_t, (const float*)W.tysum, 1.f / (float)d->N, d->momentum);
  FRX_LAUNCH_CHECK();
#define FRX_ROWS(K)                                                                                   \
  hipLaunchKernelGGL(k_head_rows<K>, dim3(d->N), dim3(256), 0, st, h, (const float*)W.cbuf, d->C,     \
                     (long)W.Cpad, labels, (const float*)W.xnorm, (const float*)W.ty,                 \
                     (const float*)state_t, (const float*)W.rowp, (float*)nullptr, (float*)nullptr, W.lse, W.rowloss, W.rowrank, part)
  FRX_KIND_SWITCH(d->kind, FRX_ROWS)
#undef FRX_ROWS
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_head_shard_rescale(int device, frx_stream_t stream, int N, const float* local_max, const float* global_max,
                                      float* part_sum) {
  FRX_CHECK_ARG(N > 0 && local_max && global_max && part_sum, "head_shard_rescale: bad args");
  FRX_ENTER(device);
  hipLaunchKernelGGL(k_shard_rescale, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N, local_max, global_max, part_sum);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_head_shard_finish(int device, frx_stream_t stream, const frx_head_desc* d, const float* state_t,
                                     const float* global_max, const float* global_sum, const float* global_rank, void* ws,
                                     size_t ws_bytes, float* norms, float* loss, float* lse, int32_t* topk) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(d->flags & 8, "head_shard_finish: the descriptor is not in class-sharded mode (flags bit 3)");
  FRX_CHECK_ARG(global_max && global_sum && global_rank && ws && loss, "head_shard_finish: NULL pointer");
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  HeadConst h = make_const(d);
#define FRX_FIN(K)                                                                                                        \
  hipLaunchKernelGGL(k_shard_finish<K>, dim3(cdiv(d->N, 256)), dim3(256), 0, st, h, d->N, (const float*)W.xnorm,           \
                     (const float*)W.ty, state_t, (const float*)W.rowp, global_max, global_sum, global_rank, W.lse, W.rowloss, W.rowrank)
  FRX_KIND_SWITCH(d->kind, FRX_FIN)
#undef FRX_FIN
  hipLaunchKernelGGL(k_head_finalize, dim3(1), dim3(256), 0, st, (const float*)W.rowloss, (const int32_t*)W.rowrank, d->N, loss,
                     topk, (const float*)W.lse, lse, (const float*)W.xnorm, norms);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

__global__ void k_head_aux(int kind, const float* __restrict__ lossg, const float* __restrict__ rowp, int N,
                           float* __restrict__ loss_g_out, float* __restrict__ rowp_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && loss_g_out) *loss_g_out = kind == FRX_MAG ? *lossg : 0.f;
  if (rowp_out && i < N) rowp_out[i] = rowp ? rowp[i] : 0.f;
}

extern "C" int frx_head_aux(int device, frx_stream_t stream, const frx_head_desc* d, const float* state_t, void* ws,
                            size_t ws_bytes, float* loss_g, float* row_param) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(ws, "head_aux: NULL workspace");
  FRX_ENTER(device);
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  const bool elastic = d->kind == FRX_ELASTIC_ARC || d->kind == FRX_ELASTIC_COS;
  FRX_CHECK_ARG(!elastic || !row_param || state_t, "%s", state_what(d->kind));
  const float* rowp = elastic ? state_t : (d->kind == FRX_ADA || d->kind == FRX_MAG) ? (const float*)W.rowp : nullptr;
  hipLaunchKernelGGL(k_head_aux, dim3(cdiv(d->N, 256)), dim3(256), 0, (hipStream_t)stream, (int)d->kind,
                     (const float*)W.lossg, rowp, d->N, loss_g, row_param);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_head_fwd(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                            const float* w, const int64_t* labels, float* state_t, void* ws,
                            size_t ws_bytes, float* cos_s, float* logits, float* norms, float* loss,
                            float* lse, int32_t* topk) {
  if (int rc = frx_head_fwd_cos(device, stream, d, x, w, labels, ws, ws_bytes, nullptr)) return rc;
  if (d->kind == FRX_VPL)
    if (int rc = frx_head_vpl_prepare(device, stream, d, x, labels, state_t, ws, ws_bytes)) return rc;
  return frx_head_fwd_loss(device, stream, d, labels, state_t, nullptr, 0, ws, ws_bytes, cos_s, logits,
                           norms, loss, lse, topk);
}

static int head_bwd_impl(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                         const float* w, const int64_t* labels, const float* state_t,
                         const float* gout, const float* dlogits, void* ws, size_t ws_bytes, float* dx,
                         float* dw, int accumulate_dw) {
  if (int rc = check_desc(d)) return rc;
  FRX_CHECK_ARG(x && w && labels && ws && dx && dw, "head_bwd: NULL pointer");
  FRX_CHECK_ARG(!needs_state_d(d) || state_t, "%s", state_what(d->kind));
  FRX_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  HeadWs W = carve(d, ws);
  if (ws_bytes < W.bytes) { set_error("head workspace too small: %zu < %zu", ws_bytes, W.bytes); return FRX_ERR_WORKSPACE; }
  HeadConst h = make_const(d);
  const float inv_n = 1.f / (float)d->N;
  const float* rowp = (d->kind == FRX_ELASTIC_ARC || d->kind == FRX_ELASTIC_COS) ? state_t : (const float*)W.rowp;
#define FRX_GRAD(K)                                                                                  \
  hipLaunchKernelGGL(k_head_grad<K>, dim3(d->N), dim3(256), 0, st, h, (const float*)W.cbuf, d->C,    \
                     W.Cpad, labels, (const float*)W.xnorm, (const float*)W.ty, state_t, rowp,       \
                     (const float*)W.lse, gout, inv_n, dlogits, W.gbuf, W.dn, W.dxh, d->D)
  FRX_KIND_SWITCH(d->kind, FRX_GRAD)
#undef FRX_GRAD
  FRX_LAUNCH_CHECK();
  const bool cd = w_is_cd(d->kind);
  const bool vpl = vpl_memory_on(d);
  if (vpl) {
    hipLaunchKernelGGL(k_vpl_split, dim3(d->N), dim3(256), 0, st, W.gbuf, W.gbuf2,
                       (const float*)(state_t + (size_t)d->C * d->D), labels, d->C, W.Cpad, d->p[0]);
    FRX_LAUNCH_CHECK();
  }
  // dX^ [N,D] = dC [N,C] . W^   (K = C: split so the grid fills the chip)
  {
    // (W.dxh was cleared by k_head_grad, one row per block -- a kernel, not hipMemsetAsync: inside a replayed hipGraph the
    // memset NODE intermittently filled this buffer with a stale 32-bit pattern instead of 0 -- seen as dfeat ~1e8..1e34
    // on a fraction of the runs, graph mode only)
    GemmArgs g{};
    g.A = W.gbuf; g.lda = W.Cpad; g.a_mcontig = 0; g.a_kscale = W.winv;
    g.B = w; g.M = d->N; g.N = d->D; g.K = d->C;
    if (cd) { g.b_ncontig = 1; g.ldb = d->D; } else { g.b_ncontig = 0; g.ldb = d->C; }
    g.C = W.dxh; g.ldc = d->D; g.atomic_out = 1;
    const int tiles = cdiv(d->N, GBM) * cdiv(d->D, GBN);
    int ksplit = tiles >= 512 ? 1 : cdiv(1024, tiles);
    ksplit = ksplit > cdiv(d->C, 64) ? cdiv(d->C, 64) : ksplit;
    if (int rc = launch_gemm(st, g, ksplit)) return rc;
    if (vpl) {                       // + (dC * a*lamda, non-target) . M^   into the same accumulator
      g.A = W.gbuf2; g.a_kscale = W.minv; g.B = state_t;
      if (int rc = launch_gemm(st, g, ksplit)) return rc;
    }
  }
  hipLaunchKernelGGL(k_norm_bwd_rows, dim3(cdiv(d->N, 4)), dim3(256), 0, st, x, (const float*)W.dxh,
                     (const float*)W.xinv, (d->kind == FRX_SPHERE || d->kind == FRX_MAG) ? (const float*)W.dn : (const float*)nullptr,
                     d->N, d->D, dx, 0);
  FRX_LAUNCH_CHECK();
  // dW^ = dC^T . X^  in the weight's own layout
  {
    GemmArgs g{};
    g.K = d->N;
    g.C = W.dwh;
    if (cd) {  // [C,D] = gbuf^T [C,N] . x^ [N,D]
      g.A = W.gbuf; g.lda = W.Cpad; g.a_mcontig = 1; g.a_kscale = W.xinv;
      g.B = x; g.ldb = d->D; g.b_ncontig = 1;
      g.M = d->C; g.N = d->D; g.ldc = d->D;
    } else {   // [D,C] = x^T [D,N] . gbuf [N,C]
      g.A = x; g.lda = d->D; g.a_mcontig = 1; g.a_kscale = W.xinv;
      g.B = W.gbuf; g.ldb = W.Cpad; g.b_ncontig = 1;
      g.M = d->D; g.N = d->C; g.ldc = d->C;
    }
    if (int rc = launch_gemm(st, g, 1)) return rc;
  }
  if (cd)
    hipLaunchKernelGGL(k_norm_bwd_rows, dim3(cdiv(d->C, 4)), dim3(256), 0, st, w, (const float*)W.dwh,
                       (const float*)W.winv, (const float*)nullptr, d->C, d->D, dw, accumulate_dw);
  else if (d->D <= NB_ROWS * NB_RL)
    hipLaunchKernelGGL(k_norm_bwd_cols_reg, dim3(cdiv(d->C, COLB)), dim3(1024), 0, st, w, (const float*)W.dwh,
                       (const float*)W.winv, d->D, d->C, dw, accumulate_dw);
  else
    hipLaunchKernelGGL(k_norm_bwd_cols, dim3(cdiv(d->C, COLB)), dim3(256), 0, st, w, (const float*)W.dwh,
                       (const float*)W.winv, d->D, d->C, dw, accumulate_dw);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_head_bwd(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                            const float* w, const int64_t* labels, const float* state_t,
                            const float* gout, void* ws, size_t ws_bytes, float* dx, float* dw,
                            int accumulate_dw) {
  return head_bwd_impl(device, stream, d, x, w, labels, state_t, gout, nullptr, ws, ws_bytes, dx, dw, accumulate_dw);
}

extern "C" int frx_head_bwd_dlogits(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                                    const float* w, const int64_t* labels, const float* state_t,
                                    const float* dlogits, void* ws, size_t ws_bytes, float* dx, float* dw,
                                    int accumulate_dw) {
  FRX_CHECK_ARG(dlogits != nullptr, "head_bwd_dlogits: dlogits is NULL");
  return head_bwd_impl(device, stream, d, x, w, labels, state_t, nullptr, dlogits, ws, ws_bytes, dx, dw, accumulate_dw);
}
