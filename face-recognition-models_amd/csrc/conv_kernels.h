// Implicit-GEMM convolution kernels for gfx950 (NHWC activations, K-contiguous weights).
//
//   forward : Y[m, co]  = sum_{r,s,ci} f(X)[pix(m,r,s), ci] * W[co][r][s][ci]       (KRSC weights)
//   dgrad   : dX[m, ci] = sum_{r,s,co} dY[pixT(m,r,s), co]  * Wt[ci][r][s][co]      (CRSK weights)
//   wgrad   : dW[co][r][s][ci] += sum_m dY[m, co] * f(X)[pix(m,r,s), ci]            (split-K, fp32 atomics)
//
// f(X) = relu(scale[c]*x + shift[c]) is the previous layer's train-mode BatchNorm + ReLU,
// applied while the tile is staged into LDS so the normalised activation never exists in
// HBM.  One K-chunk = 64 bytes of the contraction axis per row (32 bf16 / 16
// fp32), so the bf16 and fp32 instantiations share the staging code, the LDS image and
// its swizzle byte for byte; only the MFMA differs:
//   bf16: v_mfma_f32_16x16x32_bf16 on the lane's 16-byte fragment
//   fp32: 4 x v_mfma_f32_16x16x4_f32 on the 4 floats of the same 16 bytes (exact fp32 chain;
//         k order inside a chunk is permuted identically for A and B, which a sum allows)
#pragma once
#include "frx_common.h"
#include "bn_tot.h"
#include <type_traits>

namespace frx {

// Debug builds (make EXTRA=-DFRX_DBG_TIMES): thread 0 of every block records four wall-clock stamps (100 MHz)
// -- start, ring filled, K loop done, epilogue done -- read back by scripts/*_stamps.py.
#ifdef FRX_DBG_TIMES
static __device__ long long g_dbg_times[8192 * 4];
#define FRX_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_dbg_times[blockIdx.x * 4 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define FRX_DBG_EXPORT(name) extern "C" int name(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(frx::g_dbg_times), sizeof(long long) * n); }
#else
#define FRX_STAMP(k) do {} while (0)
#define FRX_DBG_EXPORT(name)
#endif

template <typename T> struct TT;
template <> struct TT<float>  { static constexpr int VEC = 4, CE = 16; };
template <> struct TT<bf16_t> { static constexpr int VEC = 8, CE = 32; };

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 as_uint4(u32x4_t v) { return make_uint4(v[0], v[1], v[2], v[3]); }

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_STEM = 2,
       MODE_FWD3 = 3, MODE_DGRAD3 = 4 };   // 3x3, stride 1, pad 1 with the gathered operand staged as a PATCH (k_igemm, "P3")
// epilogue flavours of k_igemm
enum { EPI_PLAIN = 0,       // store
       EPI_STATS = 1,       // store + per-channel sum / sum of squares (train-mode BN statistics)
       EPI_BNBWD = 2,       // dgrad: mask by e_scale*e_y+e_shift > 0, store dz, reduce sum(dz), sum(dz*xhat)
       EPI_FC = 3,          // + bias, fp32 output (the fc layer)
       EPI_BNBWD_OUT = 4 }; // as EPI_BNBWD with the merge-ReLU mask e_out > 0

struct ConvArgs {
  const void* X;          // gathered operand: activations (fwd / wgrad) or dY (dgrad)
  const void* W;          // [Ncol][R][S][Kc] (fwd: KRSC, dgrad: CRSK)
  void* Y;                // [M][Ncol]
  const float* in_scale;  // optional [Kc]: prologue affine on X ...
  const float* in_shift;
  int in_relu;            // ... followed by ReLU
  const float* bias;      // optional [Ncol]
  const void* addend;     // optional [M][Ncol] (same dtype as Y unless out_f32): Y = acc + addend
  int add_stride;         // 2: addend is the COMPACT [N,ceil(Ho/2),ceil(Wo/2),Ncol] gradient of a stride-2 1x1 branch,
                          //    added at the even pixels only (the other 3/4 of that gradient are zeros nobody stores)
  float* stat_partial;    // optional [tilesM][2][Ncol]: column sums / sums of squares of Y
  float* stat_tot;        // instead of stat_partial: replicated totals [stat_R][2][Ncol], ADDED to with float atomics (bn_tot.h)
  int stat_R;
  BnTot in_tot;           // PRO == 1 with in_scale == nullptr: the prologue's scale / shift are derived from these totals
  BnTot pro_tot;          // PRO == 2 with in_scale == nullptr: alpha / beta / gam are derived from these totals
  int out_f32;            // store Y as fp32 regardless of T
  // PRO == 2 (dgrad): the gathered operand is the BN backward  alpha*dz + beta*y + gam  of two tensors
  const void* X2;         //   y (raw conv output), same indexing as X (= dz); in_scale = alpha, in_shift = beta
  const float* pro_gam;   //   gam [Kc]
  void* dy_out;           //   optional (1x1 with PRO == 2 / 3; patch mode with any prologue): the transformed operand is also stored here, same indexing as X
  // PRO == 3 (forward, 1x1): the gathered operand is the residual MERGE of the block before,
  //   relu(in_scale*X + (id_scale*X2 + (in_shift + id_shift)))  (X = that block's raw conv3 output, X2 = its identity: the block input,
  //   id_scale = 1 / id_shift = 0, or the raw output of its projection with that BatchNorm's constants) -- what frx_block_merge_fwd
  //   computes in a pass of its own.  dy_out = the block output (stored by the first column of tiles), mask_out its > 0 bits.
  const float* id_scale; const float* id_shift;   //   [Kc] or NULL (identity); with in_tot set: derived from id_tot instead
  BnTot id_tot;           //   the projection's BatchNorm as replicated totals (tot == nullptr: plain identity)
  unsigned char* mask_out;  //   optional [M * Kc / VEC]: bit j of a byte = channel j of that 16-byte group is > 0
  // epi_bnbwd (dgrad): the output is the gradient w.r.t. a post-BN(-ReLU) activation; mask it, write dz and
  // reduce  sum(dz), sum(dz*xhat)  per channel into stat_partial (what frx_bn_bwd_reduce does in a pass of its own)
  int epi_bnbwd;
  const void* e_y;        //   raw conv output of that BN layer, [M][Ncol]
  const void* e_out;      //   optional: block output -> mask = out > 0 (merge ReLU); else mask = e_scale*y+e_shift > 0
  const unsigned char* e_bits;  // optional, instead of e_out: the same mask as one byte per 16-byte channel group (bit j =
                          //   channel j of the group is > 0), written by frx_block_merge_fwd_mask: 1/16 of the bytes
  const float* e_scale; const float* e_shift; const float* e_mean; const float* e_invstd;
  int N, Hx, Wx, Kc;      // geometry of X (Kc = its channel count)
  int Ho, Wo;             // output spatial size; M = N*Ho*Wo
  int Ncol, R, S, stride, pad;
  int M;
  int mode;
  int tilesM, tilesN;
  int nvb;                   // tiles incl. the padding of tilesM to a multiple of 8 ("virtual blocks"); the grid may be smaller (persistent blocks)
  unsigned xbytes, wbytes;   // sizes of X (and X2) and W in bytes: buffer-load bounds (out-of-range reads return 0)
  // Stride-2 input gradient by output-pixel PARITY CLASS (dgrad of a strided RxS conv): output pixel (h, w) only
  // receives taps with r = h + pad, s = w + pad (mod 2), so the four classes (h & 1, w & 1) are dense convolutions over a
  // quarter of the pixels each with 1 + 2 + 2 + 4 of the 9 taps -- 2.25 taps per pixel instead of 9 of which 3/4 gather
  // zeros.  One launch: tiles [cls_tile0[c], cls_tile0[c + 1]) belong to class c = 2 * (h & 1) + (w & 1); rows inside a
  // class run over (n, h', w') with h = 2 h' + (c >> 1), w = 2 w' + (c & 1).
  int s2c;
  int cls_tile0[5];
};

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
// 8 consecutive elements (ESZ bytes each) through a buffer descriptor, as floats; out-of-range -> 0
template <int ESZ>
__device__ __forceinline__ void buf_load8(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, int soff, float (&o)[8]) {
  if constexpr (ESZ == 4) {
    const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, soff, 0);
    const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off + 16u, soff, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = __uint_as_float(a[e]); o[4 + e] = __uint_as_float(b[e]); }
  } else {
    u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, soff, 0);
    const bf16x8 h = *reinterpret_cast<bf16x8*>(&a);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)h[e];
  }
}

// The same 8 elements kept as loaded (4 or 8 dwords) until unpack(): holds the registers of an in-flight load, not of 8 floats
template <int ESZ> struct Raw8 {
  u32x4_t a, b;      // (b unused for 2-byte elements)
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, int soff) {
    a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, soff, 0);
    if constexpr (ESZ == 4) b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off + 16u, soff, 0);
  }
  __device__ __forceinline__ void unpack(float (&o)[8]) const {
    if constexpr (ESZ == 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = __uint_as_float(a[e]); o[4 + e] = __uint_as_float(b[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(a[e] << 16); o[2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u); }
    }
  }
};

// ---- LDS-DMA (buffer_load_dwordx4 ... lds): a tile row goes from global memory straight into LDS, no VGPR hop, no ds_write.
// Semantics measured on gfx950 (scripts/probes/dma_probe.hip): the wave's 64 x 16 bytes land LANE-LINEAR at M0 + lane * 16;
// an out-of-range lane (offset past the descriptor) writes ZEROS; the scalar offset moves the global side only; the
// instruction's immediate offset moves BOTH sides (so it is not used).  hipcc knows nothing about what an asm statement
// has in flight: the kernel counts its own vmcnt (s_waitcnt below) -- with the builtin form it would wait vmcnt(0)
// before every ds_read that may alias the destination.
__device__ __forceinline__ u32x4_t raw_rsrc(const void* p, unsigned bytes) {
  const unsigned long long q = (unsigned long long)p;
  u32x4_t r;
  r[0] = (unsigned)q; r[1] = (unsigned)(q >> 32); r[2] = bytes; r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void dma16(u32x4_t rsrc, unsigned lds_addr, unsigned voff, int soff) {
  unsigned keep;      // M0 is the compiler's: save / restore it inside the one statement that uses it
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// ds_read_b128 of a [rows][64 B] image is 2-way bank-conflicted for the 16x16x32 fragment
// pattern (row = lane&15, chunk = lane>>4); XOR-ing the 16-byte chunk index with h[(row>>2)&3],
// h = {0,2,3,1}, makes every 16-lane service group hit 16 distinct slots.
__device__ __forceinline__ int swz64(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// Same for [rows][128 B] images (K-chunks of 128 bytes, 8 slots per row, a 256-byte bank row holds two rows): XOR-ing the
// slot with (row >> 1) & 7 gives the even rows -- and the odd rows -- of every 16-lane service group 8 distinct slots.
// Patch images (k_igemm P3: [rows][64 B], fragments start at ANY row): XOR-ing the slot with 2 * ((row >> 2) & 1) keeps the
// 16 consecutive rows of a fragment on 16 distinct slots per service group whatever the first row is -- of the four rows
// with equal (row & 3), one lies in each quarter of the fragment, the quarters read slots (c, c^1, c^1, c) in a group, and
// (0, 2, 0, 2) ^ every rotation of (0, 1, 1, 0) is a permutation of 0..3.  (swz64 only holds for row0 = 0 mod 16.)
__device__ __forceinline__ int pswz(int row) { return (row >> 1) & 2; }

template <int KC> __device__ __forceinline__ int swz_row(int row) {
  if constexpr (KC == 64) return swz64(row);
  else return (row >> 1) & 7;
}

// Weight-fragment row permutation: fragment j, MFMA row fr  ->  channel offset inside the wave's
// N range, chosen so fragments (2a, 2a+1) give a lane 8 consecutive channels (see the epilogue).
__device__ __forceinline__ int chan_of(int j, int fr) { return 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3); }

// Sum N values per lane over the 16 lanes of a DPP row with a halving butterfly: each step a lane keeps
// half of its values (chosen by one bit of its index) and adds the partner's copy of that half.
// Afterwards lane fr holds the 16-lane total of value index (fr % N) in v[0].  Everything is indexed
// at compile time (runtime-indexed register arrays turn into select chains or scratch).
template <int N, int BIT>
__device__ __forceinline__ void lane16_butterfly(float* vs, float* vq, int fr) {
  if constexpr (BIT >= 1) {
    if constexpr (N > BIT) {
      constexpr int H = N / 2;
      const bool up = (fr & BIT) != 0;
#pragma unroll
      for (int e = 0; e < H; ++e) {
        const float ks = up ? vs[e + H] : vs[e], ss = up ? vs[e] : vs[e + H];
        const float kq = up ? vq[e + H] : vq[e], sq = up ? vq[e] : vq[e + H];
        vs[e] = ks + __shfl_xor(ss, BIT, 64);
        vq[e] = kq + __shfl_xor(sq, BIT, 64);
      }
      lane16_butterfly<H, BIT / 2>(vs, vq, fr);
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) { vs[e] += __shfl_xor(vs[e], BIT, 64); vq[e] += __shfl_xor(vq[e], BIT, 64); }
      lane16_butterfly<N, BIT / 2>(vs, vq, fr);
    }
  }
}

template <typename T>
__device__ __forceinline__ uint4 bn_relu_vec(uint4 raw, const float* __restrict__ sc,
                                             const float* __restrict__ sh, int relu) {
  if constexpr (sizeof(T) == 4) {
    const float4 s = *reinterpret_cast<const float4*>(sc);
    const float4 b = *reinterpret_cast<const float4*>(sh);
    float4 v = *reinterpret_cast<float4*>(&raw);
    v.x = fmaf(v.x, s.x, b.x); v.y = fmaf(v.y, s.y, b.y); v.z = fmaf(v.z, s.z, b.z); v.w = fmaf(v.w, s.w, b.w);
    const float lo = relu ? 0.f : -INFINITY;      // (no per-element select on the flag)
    v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
    return *reinterpret_cast<uint4*>(&v);
  } else {
    bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
    const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sh), b1 = *reinterpret_cast<const float4*>(sh + 4);
    const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)fmaf((float)v[j], ss[j], bb[j]);
    // ReLU on the PACKED result: a bf16 is negative exactly when its bit pattern is a negative int16, so one
    // v_pk_max_i16 against 0 clamps two elements (against INT16_MIN it is the identity: no select on `relu`).
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    const short thr = relu ? (short)0 : (short)-32768;
    const s16x2_t t2 = {thr, thr};
    uint4 r = *reinterpret_cast<uint4*>(&o);
    unsigned* rw = reinterpret_cast<unsigned*>(&r);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s16x2_t x = *reinterpret_cast<s16x2_t*>(&rw[q]);
      x = __builtin_elementwise_max(x, t2);
      rw[q] = *reinterpret_cast<unsigned*>(&x);
    }
    return r;
  }
}

// alpha*dz + beta*y + gam on a 16-byte vector of each (BN backward as a prologue)
template <typename T>
__device__ __forceinline__ uint4 affine2_vec(uint4 rdz, uint4 ry, const float* __restrict__ al,
                                             const float* __restrict__ be, const float* __restrict__ ga) {
  constexpr int V = 16 / sizeof(T);
  float a_[V], b_[V], g_[V];
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    const float4 x = *reinterpret_cast<const float4*>(al + 4 * q), y = *reinterpret_cast<const float4*>(be + 4 * q),
                 z = *reinterpret_cast<const float4*>(ga + 4 * q);
    a_[4 * q] = x.x; a_[4 * q + 1] = x.y; a_[4 * q + 2] = x.z; a_[4 * q + 3] = x.w;
    b_[4 * q] = y.x; b_[4 * q + 1] = y.y; b_[4 * q + 2] = y.z; b_[4 * q + 3] = y.w;
    g_[4 * q] = z.x; g_[4 * q + 1] = z.y; g_[4 * q + 2] = z.z; g_[4 * q + 3] = z.w;
  }
  if constexpr (sizeof(T) == 4) {
    const float* d = reinterpret_cast<const float*>(&rdz);
    const float* yy = reinterpret_cast<const float*>(&ry);
    float4 o;
    o.x = fmaf(a_[0], d[0], fmaf(b_[0], yy[0], g_[0])); o.y = fmaf(a_[1], d[1], fmaf(b_[1], yy[1], g_[1]));
    o.z = fmaf(a_[2], d[2], fmaf(b_[2], yy[2], g_[2])); o.w = fmaf(a_[3], d[3], fmaf(b_[3], yy[3], g_[3]));
    return *reinterpret_cast<uint4*>(&o);
  } else {
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(&rdz);
    const bf16x8 yy = *reinterpret_cast<const bf16x8*>(&ry);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)fmaf(a_[j], (float)d[j], fmaf(b_[j], (float)yy[j], g_[j]));
    return *reinterpret_cast<uint4*>(&o);
  }
}

// relu(s3*y + (sd*idn + (b3 + bd))) on a 16-byte vector of each -- k_merge_fwd's expression (bn.hip), fma for fma: the block
// output a fused consumer stages equals the one the stand-alone merge pass would have written, bit for bit.  It is
// affine2_vec (the BN-backward prologue's  al*a + (be*b + ga)) followed by the ReLU.  `bits`: the > 0 mask of the ROUNDED
// result (bit j = element j), the backward's merge-ReLU mask.
template <typename T>
__device__ __forceinline__ uint4 merge_vec(uint4 ry, uint4 rid, const float* __restrict__ s3, const float* __restrict__ sd,
                                           const float* __restrict__ bsum, unsigned& bits) {
  uint4 v = affine2_vec<T>(ry, rid, s3, sd, bsum);
  bits = 0;
  if constexpr (sizeof(T) == 4) {
    float* o = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = fmaxf(o[j], 0.f);
      bits |= (o[j] > 0.f ? 1u : 0u) << j;
    }
  } else {
    // packed: a bf16 is negative exactly when its bit pattern is a negative int16 (v_pk_max_i16 against 0 is the ReLU of
    // two elements; fmaxf never lets a NaN through, so "> 0" is "positive, non-zero int16")
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s16x2_t x = *reinterpret_cast<s16x2_t*>(&w[q]);
      const s16x2_t z = {0, 0};
      x = __builtin_elementwise_max(x, z);
      w[q] = *reinterpret_cast<unsigned*>(&x);
      bits |= ((w[q] & 0xffffu) ? 1u : 0u) << (2 * q);
      bits |= ((w[q] >> 16) ? 1u : 0u) << (2 * q + 1);
    }
  }
  return v;
}

template <typename T> __device__ __forceinline__ float load_as_float(const void* p, long i) {
  if constexpr (sizeof(T) == 4) return reinterpret_cast<const float*>(p)[i];
  else return (float)reinterpret_cast<const bf16_t*>(p)[i];
}

// ------------------------------------------------------------------------------------------
// forward / dgrad / stem: BM x BN output tile, 4 waves as WM x WN, double-buffered LDS,
// one barrier per K-chunk, global loads of chunk k+1 in flight under the MFMAs of chunk k.
// ------------------------------------------------------------------------------------------
// dynamic LDS bytes of a k_igemm launch: the prologue's per-channel tables (scale, shift[, gam]) over the Kc input channels
static inline unsigned igemm_pro_lds(int pro, int Kc) { return pro ? (pro >= 2 ? 3u : 2u) * (unsigned)Kc * 4u : 0u; }

// MODE (gather geometry) and PRO (BN+ReLU prologue) are compile-time so the steady-state K loop is
// straight-line code: hipcc then keeps counted s_waitcnt vmcnt(N) for the register ring.
template <typename T, int BM, int BN, int WM, int WN, int MODE, int PRO, int EPI, bool ADD, int KC = 64, int PD = 3, int NS = 0, bool PERSIST = false, bool SPEC = false>   // PRO: 0 none, 1 BN+ReLU, 2 BN-backward affine, 3 residual merge of the block before (two tensors, 1x1 forward); KC: bytes of the contraction axis per row and K-chunk; PD: ring depth; NS > 0: LDS-DMA staging into a ring of NS LDS stages (PRO == 0 only) instead of the register ring
// (patch mode: three blocks per CU on the 64-column tile -- except its BN-backward-prologue input gradient, which spilled 13
// registers at 168 and must not: scratch accesses count in the hand-counted vmcnt; csrc/check_spills.py fails the build on any)
// (the merge prologue, PRO == 3, on the four-wave tiles: 168 registers without a vector spill -- three blocks per CU)
// waves per SIMD the register budget must allow: 3 where the kernel fits 168 registers without spilling (measured:
// +10-20 % on the prologue-free variants), 2 for the BN-prologue variants (they spill 35-50 registers at 3)
// (WM x WN = 4 waves; or 8 waves on a 128x128 tile for launches with too few tiles to give every SIMD two waves)
#ifndef FRX_OCC4W            // tuning aid: blocks per CU the four-wave tiles are compiled for (0: the table below)
#define FRX_OCC4W 0
#endif
__global__ __launch_bounds__(64 * WM * WN * (SPEC ? 2 : 1), SPEC ? 1 : (MODE == MODE_FWD3 || MODE == MODE_DGRAD3) ? ((BN == 64 && !(MODE == MODE_DGRAD3 && PRO == 2)) ? 3 : 2) : (NS > 4) ? (WM * WN) / 4 : (KC == 128 || PD > 3) ? 2 : (WM * WN == 8 ? 4 : (FRX_OCC4W ? FRX_OCC4W : (((PRO == 0 || (PRO == 3 && !PERSIST)) && EPI != EPI_BNBWD_OUT && !ADD) ? 3 : 2)))) void k_igemm(ConvArgs ka) {
  // P3 (MODE_FWD3 / MODE_DGRAD3): 3x3, stride 1, pad 1.  The rows a tile gathers over its nine taps are the CONTIGUOUS pixel
  // range [m0 - W - 1, m0 + BM + W + 1) of the flattened (n, h, w) axis, so per 64-byte channel chunk that range is staged
  // ONCE as a patch (LDS-DMA, then the BN prologue in place on the staging thread's own 16-byte pieces) and the nine taps
  // read their MFMA fragments out of it at per-lane row offsets computed once per tile; a tap that leaves the image points
  // at a row of zeros.  The prologue's vector-ALU work and the activation traffic drop ~8x against gathering every tap as
  // a chunk of its own (there: 49 vector instructions per 8 MFMAs, no MFMA / VALU co-execution -- profiles/r03_pmc_3x3_*);
  // the weights stream through an NS-stage LDS-DMA ring, one 64-byte chunk per tap and step.
  constexpr bool P3 = (MODE == MODE_FWD3 || MODE == MODE_DGRAD3);
  constexpr bool DMA = NS > 0 && !P3;
  static_assert(!DMA || (PRO == 0 && MODE != MODE_STEM && NS >= 3 && NS <= 8), "LDS-DMA staging: prologue-free launches, 3 to 8 stages");
  static_assert(!P3 || (KC == 64 && NS >= 4 && NS <= 8 && !PERSIST && sizeof(T) == 2), "patch mode: bf16, 64-byte chunks, 4 to 8 weight stages");
  // SPEC (patch mode, launches with at most one block per CU): WM x WN MORE waves per block that do nothing but stage -- weight
  // DMA, patch DMA, prologue pass, side-output stores -- while the first WM x WN only read fragments and issue MFMAs.  A lone
  // wave per SIMD issues one instruction per 4 clocks and ran a step's parts back to back (MFMA 256-295 + weight DMA 140 +
  // fragment reads 140 + patch 140 clocks, profiles/r03_p3_ablation_stamps.txt); two waves per SIMD with different jobs overlap.
  static_assert(!SPEC || P3, "staging waves exist in patch mode only");
  constexpr int VEC = TT<T>::VEC, CE = KC / (int)sizeof(T);      // elements per 16-byte load; elements per K-chunk
  constexpr int CPR = KC / 16;                                   // 16-byte slots per row of the LDS image
  constexpr int NT = 64 * WM * WN, RPP = NT / CPR;     // threads; tile rows staged per pass (CPR x 16-byte loads per row)
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
  constexpr int ALD = BM / RPP, BLD = BN / RPP;
  constexpr int STAGE = (BM + BN) * KC;
  static_assert(KC == 64 || KC == 128, "K-chunk of 64 or 128 bytes");
  static_assert(MODE != MODE_STEM || KC == 64, "the stem's rows are 64 bytes per tap row");
  static_assert((WM * WN == 4 || WM * WN == 8) && WTM % 16 == 0 && WTN % 32 == 0 && ALD >= 1 && BLD >= 1, "bad wave tiling");
  constexpr int PROWS = BM + 64;            // P3: rows of a patch buffer (the last one is the row of zeros): W <= 30
  constexpr int PBUF = PROWS * 64;          //     bytes of a patch buffer; [2 (+1: raw y of PRO == 2)] of them, then NS weight stages of BN rows
  __shared__ __attribute__((aligned(16))) char smem[P3 ? (PRO == 2 ? 3 : 2) * PBUF + NS * BN * 64 : (DMA ? NS : 2) * STAGE];
  static_assert(!DMA || RPP % 16 == 0, "the source-side swizzle must be the same for every row a thread stages");
  // BN scale/shift of the input channels live in LDS: fetching them from global memory at commit
  // time would be the NEWEST vector-memory op and force vmcnt(0), draining the prefetch ring.
  // (dynamic LDS, igemm_pro_lds() bytes: sized by the layer's channel count -- a fixed 2048-channel table cost 24 KB,
  // i.e. a resident block per CU, on the short-K layers.  Layout [channel / VEC][table][VEC]: the NTAB vectors of one
  // 16-byte channel group sit next to each other, so one address register with immediate offsets reaches all of them,
  // as the fixed-size tables allowed.)
  extern __shared__ __attribute__((aligned(16))) float s_pro[];
  constexpr int NTAB = PRO >= 2 ? 3 : 2;
  static_assert(PRO != 3 || (MODE == MODE_FWD && NS == 0), "the merge prologue: pointwise forward on the register ring");

  FRX_STAMP(0);
  constexpr bool STATS_ = (EPI == EPI_STATS || EPI == EPI_BNBWD || EPI == EPI_BNBWD_OUT);
  __shared__ float red[STATS_ ? 2 * WM * BN : 1];             // [2][WM][BN]: the statistics epilogue's cross-wave reduction
  // PERSISTENT form (template flag PERSIST): the grid may be smaller than the number of tiles (a.nvb "virtual blocks"); block b then runs the
  // tiles b, b + grid, b + 2 grid, ... one after the other.  (grid / 8) is a multiple of tilesN, so all of a block's tiles
  // lie in ONE column of tiles: the prologue tables are filled once per block, and the per-channel statistics of its tiles
  // accumulate in a register per thread and reach memory as ONE burst of atomics per block (bn_tot.h) instead of one per tile.
  const int tid_ = threadIdx.x;
  bool first_tile = true;
  float stat_run = 0.f;            // thread t < 2 BN: running sum of statistic (t / BN) of column (t % BN) over this block's tiles
  int n0_blk = 0;
  // (`a` is the argument block: the kernel argument itself, or -- persistent form -- a freshly laundered view of the
  // kernarg segment, so that every field is loaded where the tile uses it instead of being hoisted out of the tile loop)
  auto run_tile = [&](const auto& a, const int bid) {
  // XCD-aware tile order: blocks that share an A row-panel (same mt) share an XCD's L2.
  const int xcd = bid & 7, local = bid >> 3;
  const int mt = (local / a.tilesN) * 8 + xcd, nt = local % a.tilesN;
  if (mt >= a.tilesM) return;
  // parity-class mode (a.s2c): this tile's class, the class's pixel grid and row count (all block-uniform)
  int cls = 0, pa = 0, pb = 0, Hc = a.Ho, Wc = a.Wo, Mc = a.M, mtl = mt;
  if (MODE == MODE_DGRAD && a.s2c) {
    cls = (mt >= a.cls_tile0[1]) + (mt >= a.cls_tile0[2]) + (mt >= a.cls_tile0[3]);
    pa = cls >> 1; pb = cls & 1;
    Hc = (a.Ho - pa + 1) >> 1; Wc = (a.Wo - pb + 1) >> 1;
    Mc = a.N * Hc * Wc;
    mtl = mt - a.cls_tile0[cls];
  }
  const bool s2c = MODE == MODE_DGRAD && a.s2c;
  const int m0 = mtl * BM, n0 = nt * BN;

  int tid_l = threadIdx.x;
  if constexpr (PERSIST) asm volatile("" : "+v"(tid_l));      // (per-thread constants are re-derived per tile, not kept live across the epilogue)
  const int tid = tid_l, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const bool loader = SPEC && __builtin_amdgcn_readfirstlane(wave) >= WM * WN;      // (wave-uniform) a staging wave
  const int stid = SPEC ? (tid & (NT - 1)) : tid;                                   // thread index among the staging threads
  const int chunk = stid & (CPR - 1), srow = stid / CPR;
  // LDS-DMA writes lane-linear, so the XOR swizzle of the LDS image moves to the SOURCE side: the thread whose bytes land in
  // slot `chunk` of its row fetches the logical chunk that belongs there (the same involution the fragment reads apply)
  const int lchunk = (DMA || P3) ? (chunk ^ swz_row<KC>(srow)) : chunk;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(a.W);
  const int Ktot = a.R * a.S * a.Kc;
  const int tr0 = s2c ? ((pa + a.pad) & 1) : 0, ts0 = s2c ? ((pb + a.pad) & 1) : 0;      // first tap of the class, step 2
  const int tstep = s2c ? 2 : 1;
  const int ntaps = s2c ? ((a.R - tr0 + 1) >> 1) * ((a.S - ts0 + 1) >> 1) : a.R * a.S;
  const int nk = (MODE == MODE_STEM) ? (a.R * 32) / CE : (ntaps * a.Kc) / CE;
  const int ldw = (MODE == MODE_STEM) ? a.R * 32 : Ktot;   // weight row length in elements

  // ---- per-thread gather bookkeeping (rows fixed across the K loop)
  const bool pw = MODE != MODE_STEM && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0;   // no index arithmetic needed
  int rn[ALD], rh[ALD], rw[ALD];
  bool rok[ALD];
#pragma unroll
  for (int i = 0; i < ALD; ++i) {
    const int m = m0 + srow + RPP * i;
    rok[i] = m < Mc;
    const int mm = rok[i] ? m : 0;
    int n, oh, ow;
    if (pw) {                                                    // pointwise: the gathered pixel IS pixel m
      n = 0; oh = 0; ow = mm;
    } else {
      const int hw = Hc * Wc;
      n = mm / hw;
      const int rem = mm - n * hw;
      oh = rem / Wc; ow = rem - oh * Wc;
      if (s2c) { oh = 2 * oh + pa; ow = 2 * ow + pb; }
    }
    rn[i] = n;
    if (pw) { rh[i] = 0; rw[i] = mm; }
    else if (MODE == MODE_FWD) { rh[i] = oh * a.stride - a.pad; rw[i] = ow * a.stride - a.pad; }
    else if (MODE == MODE_DGRAD) { rh[i] = oh + a.pad; rw[i] = ow + a.pad; }
    else { rh[i] = oh * 2; rw[i] = ow * 2; }
  }
  // All tile loads are buffer loads: a 32-bit per-lane byte offset (fixed per tap) plus a SCALAR offset
  // that walks the contraction axis, so the K loop spends no vector ALU on addresses, and an offset past
  // the end of the tensor returns zeros (padding taps and tile tails need no select).
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(PRO >= 2 ? a.X2 : a.X), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, a.wbytes, 0x00020000);
  // PRO == 2 side output: the first column of tiles stores the transformed operand (dy).  The store is
  // unconditional (a branch in the K loop would cost the counted waits); every other block gets an EMPTY
  // descriptor, which drops the stores.
  // (patch mode: also the forward's transformed operand, relu(bn(x)), for the 3x3 weight gradient)
  constexpr bool SIDE = PRO == 2 || PRO == 3 || (PRO == 1 && (MODE == MODE_FWD3 || MODE == MODE_DGRAD3));
  const __amdgpu_buffer_rsrc_t rsrcDy = __builtin_amdgcn_make_buffer_rsrc(
      (SIDE && a.dy_out) ? a.dy_out : const_cast<void*>(a.X), 0, (SIDE && a.dy_out && nt == 0) ? a.xbytes : 0u, 0x00020000);
  // (PRO == 3) the merge's > 0 bits, one byte per 16-byte channel group, by the same column of tiles
  const __amdgpu_buffer_rsrc_t rsrcMask = __builtin_amdgcn_make_buffer_rsrc(
      (PRO == 3 && a.mask_out) ? (void*)a.mask_out : const_cast<void*>(a.X), 0, (PRO == 3 && a.mask_out && nt == 0) ? a.xbytes / 16u : 0u, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  unsigned bvoff[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int n = n0 + srow + RPP * i;                // rows past Ncol read zeros; their outputs are never stored
    bvoff[i] = n < a.Ncol ? (unsigned)((n * ldw + lchunk * VEC) * (int)sizeof(T)) : OOB;
  }

  // The prologue's per-channel tables (dynamic LDS).  Filled AFTER the first K-chunks' global loads have been issued: the
  // table's own loads (and, with replicated totals, the R rows per channel and the closing arithmetic) then run under the
  // latency of the tile loads instead of in front of them.
  auto fill_pro_tables = [&]() {
  if constexpr (PRO != 0) {
      if (a.in_scale) {
        for (int c = tid; c < a.Kc; c += NT) {
          float* t = s_pro + (c / VEC) * (NTAB * VEC) + (c % VEC);
          t[0] = a.in_scale[c]; t[VEC] = a.in_shift[c];
          if constexpr (PRO == 2) t[2 * VEC] = a.pro_gam[c];
          if constexpr (PRO == 3) { t[VEC] = a.id_scale ? a.id_scale[c] : 1.f; t[2 * VEC] = a.in_shift[c] + (a.id_shift ? a.id_shift[c] : 0.f); }
        }
      } else if constexpr (PRO == 3) {      // bn3's totals (and the projection BatchNorm's, if any) -> the merge's three constants
        const BnTot b = bn_tot_copy(a.in_tot);
        bn_tot_foreach<NT>(b.tot, b.R, a.Kc, [&](int c, double sm, double sq) {
          float mean, invstd, sc, sh; double var;
          bn_fwd_consts(sm, sq, b.inv_count, b.gamma[c], b.beta[c], b.eps, mean, invstd, sc, sh, var);
          float* t = s_pro + (c / VEC) * (NTAB * VEC) + (c % VEC);
          t[0] = sc; t[VEC] = 1.f; t[2 * VEC] = sh + 0.f;
        });
        if (a.id_tot.tot) {      // (the same thread owns channel c in both passes: no barrier in between)
          const BnTot bd = bn_tot_copy(a.id_tot);
          bn_tot_foreach<NT>(bd.tot, bd.R, a.Kc, [&](int c, double sm, double sq) {
            float mean, invstd, sc, sh; double var;
            bn_fwd_consts(sm, sq, bd.inv_count, bd.gamma[c], bd.beta[c], bd.eps, mean, invstd, sc, sh, var);
            float* t = s_pro + (c / VEC) * (NTAB * VEC) + (c % VEC);
            t[VEC] = sc; t[2 * VEC] = t[2 * VEC] + sh;
          });
        }
      } else if constexpr (PRO == 1) {      // the producer's replicated totals -> scale / shift (bn_tot.h)
        const BnTot b = bn_tot_copy(a.in_tot);
        bn_tot_foreach<NT>(b.tot, b.R, a.Kc, [&](int c, double sm, double sq) {
          float mean, invstd, sc, sh; double var;
          bn_fwd_consts(sm, sq, b.inv_count, b.gamma[c], b.beta[c], b.eps, mean, invstd, sc, sh, var);
          float* t = s_pro + (c / VEC) * (NTAB * VEC) + (c % VEC);
          t[0] = sc; t[VEC] = sh;
        });
      } else {                              // (sum dz, sum dz * xhat) -> alpha / beta / gam
        const BnTot b = bn_tot_copy(a.pro_tot);
        bn_tot_foreach<NT>(b.tot, b.R, a.Kc, [&](int c, double sa, double sb) {
          float al, be, ga;
          bn_bwd_consts(sa, sb, b.inv_count, b.gamma[c], b.mean[c], b.invstd[c], al, be, ga);
          float* t = s_pro + (c / VEC) * (NTAB * VEC) + (c % VEC);
          t[0] = al; t[VEC] = be; t[2 * VEC] = ga;
        });
      }
      __syncthreads();
    }
  };
  int tr = tr0, ts = ts0, c0 = 0;   // current tap (r, s) and channel offset of the K-chunk
  // Register ring of PD K-chunks: HBM/L2 latency (~2k cycles under load) is several chunks of MFMA work,
  // so loads run PD-1 chunks ahead of the LDS write that consumes them.
  constexpr int RD = DMA ? 1 : PD;
  uint4 ra[RD][ALD], rb[RD][BLD];
  uint4 ra2[PRO >= 2 ? PD : 1][ALD];   // second gathered tensor (PRO == 2: raw y; PRO == 3: the identity)
  int rc0[RD];                  // channel offset each ring slot was loaded at (for the BN prologue)
  unsigned rmask[RD];           // which of the slot's A rows were in bounds (padding stays exactly 0)
  unsigned avoff[ALD];          // byte offset of the gathered pixel of the CURRENT tap (OOB when out of bounds)
  bool aok[ALD];

  // Address arithmetic and bounds checks run once per tap, not once per K-chunk: inside a tap the
  // gather only walks along the channel axis.
  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
      bool ok = rok[i];
      int off;
      if (pw) {
        off = rw[i] * a.Kc;
      } else if (MODE == MODE_FWD) {
        const int hi = rh[i] + tr, wi = rw[i] + ts;
        ok = ok && (unsigned)hi < (unsigned)a.Hx && (unsigned)wi < (unsigned)a.Wx;
        off = ((rn[i] * a.Hx + hi) * a.Wx + wi) * a.Kc;
      } else if (MODE == MODE_DGRAD) {
        const int th = rh[i] - tr, tw = rw[i] - ts;
        const int sm = a.stride - 1, sh = a.stride >> 1;   // stride is 1 or 2
        ok = ok && th >= 0 && tw >= 0 && ((th & sm) == 0) && ((tw & sm) == 0);
        const int hi = th >> sh, wi = tw >> sh;
        ok = ok && hi < a.Hx && wi < a.Wx;
        off = ((rn[i] * a.Hx + hi) * a.Wx + wi) * a.Kc;
      } else {  // stem: physically padded NHWC4 input, row tr, 8 taps x 4 channels = 32 elements
        off = ((rn[i] * a.Hx + rh[i] + tr) * a.Wx + rw[i]) * 4;
      }
      aok[i] = ok;
      avoff[i] = ok ? (unsigned)((off + lchunk * VEC) * (int)sizeof(T)) : OOB;
    }
  };
  if constexpr (!P3) set_tap();

  // issue the global loads of chunk kc into ring slot `slot` (no transform yet: nothing waits here)
  auto issue_chunk = [&](int kc, auto slot_tag) {
    constexpr int slot = decltype(slot_tag)::value;
    rc0[slot] = c0;
    const int so = c0 * (int)sizeof(T);             // scalar offsets: wave-uniform
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
      ra[slot][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX, avoff[i], so, 0));
      if constexpr (PRO >= 2) ra2[slot][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX2, avoff[i], so, 0));
      m |= (aok[i] ? 1u : 0u) << i;
    }
    rmask[slot] = m;
    const int sob = s2c ? ((tr * a.S + ts) * a.Kc + c0) * (int)sizeof(T) : kc * KC;   // (the class walks a subset of the taps)
#pragma unroll
    for (int i = 0; i < BLD; ++i) rb[slot][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcW, bvoff[i], sob, 0));
    // advance the tap walker to chunk kc+1
    c0 += CE;
    const int span = (MODE == MODE_STEM) ? 32 : a.Kc;
    if (c0 >= span) {
      c0 = 0;
      if (MODE == MODE_STEM) { ++tr; } else { ts += tstep; if (ts >= a.S) { ts = ts0; tr += tstep; } }
      if (kc + 1 < nk) set_tap();
    }
  };
  // LDS-DMA form of issue + commit: chunk kc goes straight into LDS stage `stage` (wave w's 64 lanes fill 1 KiB = the
  // 64 / CPR tile rows starting at row w * 64 / CPR + RPP * i, lane-linear); nothing waits here
  const u32x4_t rawX = raw_rsrc(a.X, a.xbytes), rawW = raw_rsrc(a.W, a.wbytes);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  const unsigned wrow = (unsigned)__builtin_amdgcn_readfirstlane(SPEC ? (wave & (WM * WN - 1)) : wave) * 1024u;     // this wave's slice of every RPP-row pass
  auto dma_chunk = [&](int kc, int stage) {
    const int so = c0 * (int)sizeof(T);
    const unsigned sa = lds0 + (unsigned)stage * STAGE + wrow, sb = sa + BM * KC;
#pragma unroll
    for (int i = 0; i < ALD; ++i) dma16(rawX, sa + i * RPP * KC, avoff[i], so);
    const int sob = s2c ? ((tr * a.S + ts) * a.Kc + c0) * (int)sizeof(T) : kc * KC;
#pragma unroll
    for (int i = 0; i < BLD; ++i) dma16(rawW, sb + i * RPP * KC, bvoff[i], sob);
    c0 += CE;
    if (c0 >= a.Kc) {
      c0 = 0;
      ts += tstep; if (ts >= a.S) { ts = ts0; tr += tstep; }
      if (kc + 1 < nk) set_tap();
    }
  };
  // BN+ReLU prologue, then registers -> LDS stage `buf`
  auto commit_chunk = [&](int buf, auto slot_tag) {
    constexpr int slot = decltype(slot_tag)::value;
    char* As = smem + buf * STAGE;
    char* Bs = As + BM * KC;
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
      uint4 v = ra[slot][i];
      if constexpr (PRO == 1)
        v = bn_relu_vec<T>(v, s_pro + (rc0[slot] + chunk * VEC) * NTAB, s_pro + (rc0[slot] + chunk * VEC) * NTAB + VEC, a.in_relu);
      if constexpr (PRO == 2)
        v = affine2_vec<T>(v, ra2[slot][i], s_pro + (rc0[slot] + chunk * VEC) * NTAB, s_pro + (rc0[slot] + chunk * VEC) * NTAB + VEC,
                           s_pro + (rc0[slot] + chunk * VEC) * NTAB + 2 * VEC);
      unsigned mbits = 0;
      if constexpr (PRO == 3) {
        const float* tb = s_pro + (rc0[slot] + chunk * VEC) * NTAB;
        v = merge_vec<T>(v, ra2[slot][i], tb, tb + VEC, tb + 2 * VEC, mbits);
      }
      if constexpr (PRO != 0) {      // out-of-range loads are already 0; a prologue would turn them into f(0)
        if (!((rmask[slot] >> i) & 1u)) v = make_uint4(0, 0, 0, 0);
      }
      if constexpr (PRO == 2 || PRO == 3) {      // (soffset stays the literal 0: see the note on stores in the epilogue)
        u32x4_t sv; sv[0] = v.x; sv[1] = v.y; sv[2] = v.z; sv[3] = v.w;
        const unsigned so_b = avoff[i] + (unsigned)(rc0[slot] * (int)sizeof(T));
        __builtin_amdgcn_raw_buffer_store_b128(sv, rsrcDy, so_b, 0, 0);
        if constexpr (PRO == 3) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)mbits, rsrcMask, so_b >> 4, 0, 0);   // (rows past M: offset past the descriptor)
      }
      const int row = srow + RPP * i;
      *reinterpret_cast<uint4*>(As + (row * CPR + (chunk ^ swz_row<KC>(row))) * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      const int row = srow + RPP * i;
      *reinterpret_cast<uint4*>(Bs + (row * CPR + (chunk ^ swz_row<KC>(row))) * 16) = rb[slot][i];
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // (LDS-DMA path) the two halves of compute_chunk, so that the fragment reads of chunk k + 1 can be in flight under the
  // MFMAs of chunk k: two fragment sets, indexed at compile time only
  constexpr int KS = KC / 64;
  uint4 pfa[(DMA || P3) ? 2 : 1][KS][FM], pfb[(DMA || P3) ? 2 : 1][KS][FN];
  auto read_frags = [&](int stage, auto par_tag) {
    constexpr int P = decltype(par_tag)::value;
    const char* As = smem + stage * STAGE;
    const char* Bs = As + BM * KC;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WTN + chan_of(j, fr);
        pfb[P][ks][j] = *reinterpret_cast<const uint4*>(Bs + (row * CPR + ((ks * 4 + fq) ^ swz_row<KC>(row))) * 16);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 16 + fr;
        pfa[P][ks][i] = *reinterpret_cast<const uint4*>(As + (row * CPR + ((ks * 4 + fq) ^ swz_row<KC>(row))) * 16);
      }
    }
  };
  auto mfma_frags = [&](auto par_tag) {
    constexpr int P = decltype(par_tag)::value;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&pfb[P][ks][j]),
                                                                *reinterpret_cast<bf16x8*>(&pfa[P][ks][i]), acc[i][j], 0, 0, 0);
          } else {
            const float* pa = reinterpret_cast<const float*>(&pfa[P][ks][i]);
            const float* pb = reinterpret_cast<const float*>(&pfb[P][ks][j]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[q], pa[q], acc[i][j], 0, 0, 0);
          }
        }
  };
  auto compute_chunk = [&](int cur) {
    const char* As = smem + cur * STAGE;
    const char* Bs = As + BM * KC;
#pragma unroll
    for (int ks = 0; ks < KC / 64; ++ks) {             // one MFMA k-step = 64 bytes of the contraction axis
      uint4 fa[FM], fb[FN];
      // (weight fragments first: LDS returns reads in issue order, and the first MFMAs need fb[0..] and fa[0] only)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WTN + chan_of(j, fr);
        fb[j] = *reinterpret_cast<const uint4*>(Bs + (row * CPR + ((ks * 4 + fq) ^ swz_row<KC>(row))) * 16);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 16 + fr;
        fa[i] = *reinterpret_cast<const uint4*>(As + (row * CPR + ((ks * 4 + fq) ^ swz_row<KC>(row))) * 16);
      }
      // Operands are swapped (weights first): D[row = channel][col = pixel], so a lane ends up with
      // 4 CONSECUTIVE channels of one pixel per fragment -- contiguous in NHWC memory.
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&fb[j]),
                                                                *reinterpret_cast<bf16x8*>(&fa[i]), acc[i][j], 0, 0, 0);
          } else {
            const float* pa = reinterpret_cast<const float*>(&fa[i]);
            const float* pb = reinterpret_cast<const float*>(&fb[j]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[q], pa[q], acc[i][j], 0, 0, 0);
          }
        }
    }
  };
  // chunk j lives in ring slot j % PD and in LDS stage j & 1.  One ring step: MFMAs of chunk k, global loads of chunk
  // k+PD into the slot chunk k just left, registers -> LDS of chunk k+1, barrier.
  // Steady state: no branch between a load's issue and its wait, so hipcc keeps counted vmcnt(N) waits.
  auto ring_step = [&](int k, auto u_tag, auto guarded_tag) {
    constexpr int U = decltype(u_tag)::value;
    constexpr bool GUARDED = decltype(guarded_tag)::value;
    compute_chunk(k & 1);
    if (!GUARDED || k + PD < nk) issue_chunk(k + PD, std::integral_constant<int, U % PD>{});
    if (!GUARDED || k + 1 < nk) commit_chunk((k + 1) & 1, std::integral_constant<int, (U + 1) % PD>{});
    __syncthreads();
  };
  auto ring_round = [&](int k0, auto guarded_tag) {       // PD consecutive steps starting at a chunk index k0 = 0 (mod PD)
    constexpr bool GUARDED = decltype(guarded_tag)::value;
    ring_step(k0, std::integral_constant<int, 0>{}, guarded_tag);
    if constexpr (PD > 1) { if (!GUARDED || k0 + 1 < nk) ring_step(k0 + 1, std::integral_constant<int, 1>{}, guarded_tag); }
    if constexpr (PD > 2) { if (!GUARDED || k0 + 2 < nk) ring_step(k0 + 2, std::integral_constant<int, 2>{}, guarded_tag); }
    if constexpr (PD > 3) { if (!GUARDED || k0 + 3 < nk) ring_step(k0 + 3, std::integral_constant<int, 3>{}, guarded_tag); }
  };
  static_assert(PD >= 2 && PD <= 4, "ring depth");
  if constexpr (P3) {
    constexpr int NPATCH = PRO == 2 ? 3 : 2;            // patch buffers (the third: raw y of the patch being staged)
    constexpr int NPP = (PROWS * 4) / NT;               // 16-byte pieces per thread and patch
    constexpr int NPL = NPP * (PRO == 2 ? 2 : 1);       // LDS-DMA instructions per thread and patch
    constexpr int BST = BN * 64;                        // bytes of a weight stage
    constexpr int TF = NS - 1 > 5 ? NS - 1 : 5;         // tap step during which the NEXT patch gets its prologue (its pieces have landed from step NS - 1 on)
    static_assert((PROWS * 4) % NT == 0 && TF <= 7 && TF + NS - 2 <= 8 + TF && BLD * RPP == BN, "patch pieces / weight rows must tile the threads");
    const int Wd = a.Wx, HWd = a.Hx * a.Wx;
    const int PR = BM + 2 * Wd + 2;                     // patch rows in use (host: <= PROWS - 1)
    const int plo = m0 - Wd - 1;                        // pixel of patch row 0
    const int ncc = a.Kc / CE;                          // channel chunks (host: even)
    // this thread's pieces of a patch: piece q = u * NT + tid is slot (q & 3) of row (q >> 2), lane-linear for the DMA
    unsigned pvoff[NPP], pstoff[NPP];
    int ptab[NPP];
    bool pfix[NPP];
#pragma unroll
    for (int u = 0; u < NPP; ++u) {
      const int q = u * NT + stid, row = q >> 2, lc = (q & 3) ^ pswz(row), pix = plo + row;
      const bool ok = row < PR && pix >= 0 && pix < a.M;
      pvoff[u] = ok ? (unsigned)((pix * a.Kc + lc * VEC) * (int)sizeof(T)) : OOB;
      ptab[u] = lc * VEC * NTAB;
      pfix[u] = row < PR;
      // side output (the transformed operand, for the weight gradient: dy, or the forward's relu(bn(x))): every pixel is the OWN row of exactly
      // one row tile (patch rows W + 1 .. W + BM); the first column of tiles stores them (rsrcDy is empty elsewhere)
      pstoff[u] = (ok && row > Wd && row <= Wd + BM) ? pvoff[u] : OOB;
    }
    constexpr int NST = PRO != 0 ? NPP : 0;             // stores per thread and patch: issued UNCONDITIONALLY (they count in vmcnt)
    const __amdgpu_buffer_rsrc_t rsrcDy0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, 0u, 0x00020000);
    // fragment row i of this lane at tap t: byte offset inside a patch buffer (the row of zeros when the tap leaves the
    // image); filled in after the tile's first loads have been issued
    unsigned aaddr[9][FM];
    const u32x4_t rawX2 = raw_rsrc(PRO == 2 ? a.X2 : a.X, a.xbytes), rawX0 = raw_rsrc(a.X, 0), rawW0 = raw_rsrc(a.W, 0);   // (..0: empty descriptors -- every lane out of range, the DMA writes zeros)
    int b_tap = 0, b_cc = 0;                            // (tap, channel chunk) of the next weight chunk to issue
    auto dma_b = [&](int stage) {                       // past the last chunk: zeros into a stage nobody reads (keeps the vmcnt arithmetic uniform)
      if constexpr (SPEC) {      // (inside the staging waves' branch hipcc otherwise treats the walkers as divergent: VGPR operands in the asm)
        b_cc = __builtin_amdgcn_readfirstlane(b_cc); b_tap = __builtin_amdgcn_readfirstlane(b_tap);
        stage = __builtin_amdgcn_readfirstlane(stage);
      }
      const u32x4_t rw = b_cc < ncc ? rawW : rawW0;
      const int sob = (b_tap * a.Kc + b_cc * CE) * (int)sizeof(T);
      const unsigned sb = lds0 + NPATCH * PBUF + (unsigned)stage * BST + wrow;
#pragma unroll
      for (int i = 0; i < BLD; ++i) dma16(rw, sb + i * RPP * 64, bvoff[i], sob);
      if (++b_tap == 9) { b_tap = 0; ++b_cc; }
    };
    auto dma_patch = [&](int cc, int buf) {
      if constexpr (SPEC) cc = __builtin_amdgcn_readfirstlane(cc);
      const bool live = cc < ncc;
      const u32x4_t rx = live ? rawX : rawX0, rx2 = live ? rawX2 : rawX0;
      const int so = cc * CE * (int)sizeof(T);
#pragma unroll
      for (int u = 0; u < NPP; ++u) {
        dma16(rx, lds0 + (unsigned)buf * PBUF + u * NT * 16 + wrow, pvoff[u], so);
        if constexpr (PRO == 2) dma16(rx2, lds0 + 2 * PBUF + u * NT * 16 + wrow, pvoff[u], so);
      }
    };
    // the prologue, in place, on the pieces this thread staged itself (its own vmcnt covers them: no barrier in between)
    auto fix_patch = [&](int cc, int buf) {
      if constexpr (PRO != 0) {
        const __amdgpu_buffer_rsrc_t rdy = cc < ncc ? rsrcDy : rsrcDy0;      // (the step after the last chunk works on zeros: nothing of it may be stored)
#pragma unroll
        for (int u = 0; u < NPP; ++u) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (pfix[u]) {
            char* pp = smem + buf * PBUF + (u * NT + stid) * 16;
            v = *reinterpret_cast<uint4*>(pp);
            const float* tb = s_pro + cc * CE * NTAB + ptab[u];
            if constexpr (PRO == 1) v = bn_relu_vec<T>(v, tb, tb + VEC, a.in_relu);
            else v = affine2_vec<T>(v, *reinterpret_cast<const uint4*>(smem + 2 * PBUF + (u * NT + stid) * 16), tb, tb + VEC, tb + 2 * VEC);
            *reinterpret_cast<uint4*>(pp) = v;
          }
          if constexpr (PRO != 0) {      // (soffset stays the literal 0: see the note on stores in the epilogue)
            u32x4_t sv; sv[0] = v.x; sv[1] = v.y; sv[2] = v.z; sv[3] = v.w;
            __builtin_amdgcn_raw_buffer_store_b128(sv, rdy, pstoff[u] + (unsigned)(cc * CE * (int)sizeof(T)), 0, 0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // in LDS before this wave signals the next (raw) barrier
      }
    };
    auto read_p3 = [&](auto tap_tag, auto buf_tag, int stage, auto par_tag) {
      constexpr int TAP = decltype(tap_tag)::value, BUF = decltype(buf_tag)::value, P = decltype(par_tag)::value;
      const char* Bs = smem + NPATCH * PBUF + stage * BST;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WTN + chan_of(j, fr);
        pfb[P][0][j] = *reinterpret_cast<const uint4*>(Bs + (row * 4 + (fq ^ swz64(row))) * 16);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) pfa[P][0][i] = *reinterpret_cast<const uint4*>(smem + BUF * PBUF + aaddr[TAP][i]);
    };
    int stg = 0;                                        // LDS stage of the current step's weight chunk (step % NS)
    // Step k = 9 cc + t (channel chunk cc, tap t):  wait until weight chunk k + 1 has landed | barrier | weight chunk
    // k + NS - 1 into the stage chunk k - 1 just left | t = 0: DMA of patch cc + 1 into the other buffer (last read two
    // barriers ago) | fragments of step k + 1 | t = TF: prologue on patch cc + 1 (visible after the next barrier, first read
    // at t = 8) | MFMAs of step k.  Straight-line per pair of chunks: every count below is a compile-time constant.
    // ROLE: 0 every wave does everything; SPEC: 1 a staging wave's share of the step, 2 a computing wave's (the two loops are
    // separate code -- a role branch inside the step would join two register assignments sixteen times per chunk pair)
    auto step = [&](auto t_tag, auto buf_tag, auto first_tag, auto role_tag, int cc) {
      constexpr int ROLE = decltype(role_tag)::value;
      constexpr int t = decltype(t_tag)::value, BUF = decltype(buf_tag)::value, P = (t + BUF) & 1;      // P = k & 1 (9 cc = cc mod 2)
      constexpr bool FIRST = decltype(first_tag)::value;                                               // chunk 0 of the tile
      const int nxt = stg + 1 == NS ? 0 : stg + 1, prv = stg == 0 ? NS - 1 : stg - 1;
      // younger than weight chunk k + 1 (vmcnt counts loads, stores and LDS-DMA together, in issue order): weight chunks
      // k + 2 .. k + NS - 2; for NS - 2 steps after tap 0 the next patch; and the side-output stores of a prologue pass for the
      // NS - 2 steps that follow it -- this chunk's (issued at t = TF), the previous chunk's where that window wraps, and in
      // chunk 0 those of the pass in front of the loop (they sit between weight chunks NS - 2 and NS - 1)
      constexpr bool ST = (t > TF && t <= TF + NS - 2) || (!FIRST && t + 9 <= TF + NS - 2) || (FIRST && t <= NS - 3);
#ifndef FRX_P3_ABL          // timing ablations (wrong results): 1 no weight DMA, 2 no fragment reads, 4 no barrier, 8 no MFMA, 16 no patch DMA / prologue
#define FRX_P3_ABL 0
#endif
      if constexpr (ROLE != 0) {
        if constexpr (ROLE == 1) {
          wait_vmcnt<(NS - 3) * BLD + ((t >= 1 && t <= NS - 2) ? NPL : 0) + (ST ? NST : 0)>();
          __builtin_amdgcn_s_barrier();
          dma_b(prv);
          if constexpr (t == 0) dma_patch(cc + 1, 1 - BUF);
          if constexpr (t == TF) fix_patch(cc + 1, 1 - BUF);
        } else {
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          read_p3(std::integral_constant<int, (t + 1) % 9>{}, std::integral_constant<int, (t == 8 ? 1 - BUF : BUF)>{}, nxt, std::integral_constant<int, 1 - P>{});
          mfma_frags(std::integral_constant<int, P>{});
#pragma unroll
          for (int q = 0; q < FM * FN; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < FM + FN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        stg = nxt;
        return;
      }
      wait_vmcnt<(NS - 3) * BLD + ((t >= 1 && t <= NS - 2) ? NPL : 0) + (ST ? NST : 0)>();
      if constexpr (!(FRX_P3_ABL & 4)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);                // (MFMAs touch registers only: without this hipcc pulls the NEXT step's up to right behind their fragment reads)
      if constexpr (!(FRX_P3_ABL & 1)) dma_b(prv);
      if constexpr (t == 0 && !(FRX_P3_ABL & 16)) dma_patch(cc + 1, 1 - BUF);
      if constexpr (!(FRX_P3_ABL & 2)) read_p3(std::integral_constant<int, (t + 1) % 9>{}, std::integral_constant<int, (t == 8 ? 1 - BUF : BUF)>{}, nxt, std::integral_constant<int, 1 - P>{});
      if constexpr (t == TF && !(FRX_P3_ABL & 16)) fix_patch(cc + 1, 1 - BUF);
      if constexpr (!(FRX_P3_ABL & 8)) mfma_frags(std::integral_constant<int, P>{});
      // One wave per SIMD issues one instruction per 4 clocks: the step's ~45 scalar / LDS / address instructions must go
      // INTO the 16-clock shadows of its 16 MFMAs, not in front of them (measured 700 clocks per step against 256 of MFMA).
#pragma unroll
      for (int q = 0; q < FM * FN; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
        if (q < FM + FN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one fragment read of the next step
        __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);                      // scalar bookkeeping of the DMAs
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                      // address arithmetic / prologue arithmetic
      }
      __builtin_amdgcn_sched_barrier(0);                // (this step's MFMAs stay in front of the next barrier, i.e. behind reads issued a step earlier)
      stg = nxt;
    };
    auto chunk9 = [&](auto buf_tag, auto first_tag, auto role_tag, int cc) {
      step(std::integral_constant<int, 0>{}, buf_tag, first_tag, role_tag, cc); step(std::integral_constant<int, 1>{}, buf_tag, first_tag, role_tag, cc);
      step(std::integral_constant<int, 2>{}, buf_tag, first_tag, role_tag, cc); step(std::integral_constant<int, 3>{}, buf_tag, first_tag, role_tag, cc);
      step(std::integral_constant<int, 4>{}, buf_tag, first_tag, role_tag, cc); step(std::integral_constant<int, 5>{}, buf_tag, first_tag, role_tag, cc);
      step(std::integral_constant<int, 6>{}, buf_tag, first_tag, role_tag, cc); step(std::integral_constant<int, 7>{}, buf_tag, first_tag, role_tag, cc);
      step(std::integral_constant<int, 8>{}, buf_tag, first_tag, role_tag, cc);
    };
    auto all_chunks = [&](auto role_tag) {
      if constexpr (NST != 0) {                         // (chunk 0 has waits of its own only where stores are counted)
        chunk9(std::integral_constant<int, 0>{}, std::true_type{}, role_tag, 0);
        chunk9(std::integral_constant<int, 1>{}, std::false_type{}, role_tag, 1);
      }
      for (int cc = NST != 0 ? 2 : 0; cc < ncc; cc += 2) {
        chunk9(std::integral_constant<int, 0>{}, std::false_type{}, role_tag, cc);
        chunk9(std::integral_constant<int, 1>{}, std::false_type{}, role_tag, cc + 1);
      }
    };
    if (tid < 8) *reinterpret_cast<uint4*>(smem + (tid >> 2) * PBUF + (PROWS - 1) * 64 + (tid & 3) * 16) = make_uint4(0, 0, 0, 0);
    if (!SPEC || loader) {
      dma_patch(0, 0);
#pragma unroll
      for (int j = 0; j < NS - 1; ++j) dma_b(j);
    }
    __builtin_amdgcn_sched_barrier(0);                  // (the tile's first loads are on their way before the arithmetic below)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int lr = wm * WTM + 16 * i + fr, pm = m0 + lr;
      const bool ok = pm < a.M;
      const int pp = ok ? pm : 0, n = pp / HWd, rem = pp - n * HWd, h = rem / Wd, w = rem - h * Wd;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dr = MODE == MODE_FWD3 ? t / 3 - 1 : 1 - t / 3, dc = MODE == MODE_FWD3 ? t % 3 - 1 : 1 - t % 3;
        const bool valid = ok && (unsigned)(h + dr) < (unsigned)a.Hx && (unsigned)(w + dc) < (unsigned)Wd;
        const int prow = lr + Wd + 1 + dr * Wd + dc;
        aaddr[t][i] = valid ? (unsigned)(prow * 64 + ((fq ^ pswz(prow)) << 4)) : (unsigned)((PROWS - 1) * 64 + (fq << 4));
      }
    }
    fill_pro_tables();
    if (!SPEC || loader) {
      wait_vmcnt<(NS - 1) * BLD>();                     // patch 0 is older than every weight chunk
      fix_patch(0, 0);
      wait_vmcnt<(NS - 2) * BLD + NST>();               // weight chunk 0 (younger: the other chunks and the pass's stores)
    }
    __syncthreads();
    if (!SPEC || !loader) read_p3(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{});
    FRX_STAMP(1);
    // (the last step still waits, syncs and reads "the fragments of step nk" -- zeros from the empty descriptors: one
    // barrier too many per tile buys a loop without a tail)
    if constexpr (SPEC) {
      if (loader) {
        all_chunks(std::integral_constant<int, 1>{});
        wait_vmcnt<0>();
        return;                                         // (the epilogue's barriers count the waves that are still alive)
      }
      all_chunks(std::integral_constant<int, 2>{});
    } else {
      all_chunks(std::integral_constant<int, 0>{});
      wait_vmcnt<0>();
    }
  } else if constexpr (DMA) {
    // Chunk j lives in LDS stage j % NS.  Software pipeline, one barrier per chunk:
    //   step k:  wait until this wave's part of chunk k + 1 has landed (counted: the younger chunks stay in flight across
    //            the barrier) | barrier: every wave's part of chunk k + 1 has landed, and every wave has consumed the
    //            fragments of chunk k - 1 | issue chunk k + NS - 1 into the stage chunk k - 1 just left | read the fragments
    //            of chunk k + 1 (they arrive under the MFMAs) | MFMAs of chunk k on the fragments read during step k - 1.
    // The stage of chunk k itself sits idle during step k (its fragments are in registers): re-using it one step
    // earlier would race the DMA against fragment reads another wave may still have in flight.
    constexpr int LPC = ALD + BLD;                    // DMA instructions per thread and chunk
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
      if (j < nk) dma_chunk(j, j);
    if (nk >= NS - 1) wait_vmcnt<(NS - 2) * LPC>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, P0{});
    FRX_STAMP(1);
    int k = 0, stg = 0;                               // stg = k % NS
    // (STEADY: no branch between the fragment reads and the MFMAs -- at a join hipcc would wait lgkmcnt(0), i.e. for the
    // reads it has just issued, instead of only for the older set the MFMAs take)
    auto pstep = [&](auto par_tag, auto steady_tag) {
      constexpr int P = decltype(par_tag)::value;
      constexpr bool STEADY = decltype(steady_tag)::value;
      const int nxt = stg + 1 == NS ? 0 : stg + 1, prv = stg == 0 ? NS - 1 : stg - 1;
      if (STEADY || k + 1 < nk) {
        // issued so far: chunks 0 .. min(nk, k + NS - 1) - 1; chunk k + 1 must have landed
        const int younger = (k + NS - 1 < nk ? k + NS - 1 : nk) - (k + 2);
        if (NS > 3 && (STEADY || younger >= NS - 3)) wait_vmcnt<(NS - 3) * LPC>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (STEADY || k + NS - 1 < nk) dma_chunk(k + NS - 1, prv);
        read_frags(nxt, std::integral_constant<int, 1 - P>{});
      }
      mfma_frags(par_tag);
      stg = nxt;
      ++k;
    };
    for (; k + NS < nk; ) {               // two steady steps: k + 1 + NS - 1 < nk
      pstep(P0{}, std::true_type{});
      pstep(P1{}, std::true_type{});
    }
    while (k < nk) {
      pstep(P0{}, std::false_type{});
      if (k >= nk) break;
      pstep(P1{}, std::false_type{});
    }
  } else {
  issue_chunk(0, std::integral_constant<int, 0>{});
  if constexpr (PD > 1) { if (1 < nk) issue_chunk(1, std::integral_constant<int, 1>{}); }
  if constexpr (PD > 2) { if (2 < nk) issue_chunk(2, std::integral_constant<int, 2>{}); }
  if constexpr (PD > 3) { if (3 < nk) issue_chunk(3, std::integral_constant<int, 3>{}); }
  if (first_tile) fill_pro_tables();
  commit_chunk(0, std::integral_constant<int, 0>{});
  __syncthreads();
  FRX_STAMP(1);
  int kc = 0;
  for (; kc + 2 * PD - 1 < nk; kc += PD) ring_round(kc, std::false_type{});
  for (; kc < nk; kc += PD) ring_round(kc, std::true_type{});        // tail (and the whole loop when K is short)
  }

  // ---- epilogue.  C/D map: col = lane&15 (pixel), row = (lane>>4)*4 + reg (channel slot).
  // Fragment pair (2a, 2a+1) holds, for this lane, channels 32a + 8*fq + [0..8) of pixel fr: 16-byte
  // buffer stores.  The epilogue flavour is a template parameter and every access is a buffer access
  // (rows past M fall outside the descriptor: loads give 0, stores are dropped), so this is
  // straight-line code without per-element predicates; Ncol % BN == 0 is checked on the host.
  FRX_STAMP(2);
  constexpr bool OUT32 = (EPI == EPI_FC) || sizeof(T) == 4;
  constexpr int OSZ = OUT32 ? 4 : 2;
  const unsigned ybytes = (unsigned)a.M * (unsigned)a.Ncol * OSZ;
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, ybytes, 0x00020000);
  const unsigned ebytes = (unsigned)a.M * (unsigned)a.Ncol * (unsigned)sizeof(T);
  const __amdgpu_buffer_rsrc_t rsrcEy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((EPI == EPI_BNBWD || EPI == EPI_BNBWD_OUT) ? a.e_y : a.Y), 0, ebytes, 0x00020000);
  const bool ebits = EPI == EPI_BNBWD_OUT && a.e_bits != nullptr;          // block-uniform
  const __amdgpu_buffer_rsrc_t rsrcEo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(EPI == EPI_BNBWD_OUT ? (ebits ? (const void*)a.e_bits : a.e_out) : a.Y), 0,
      ebits ? (unsigned)a.M * (unsigned)a.Ncol / (unsigned)VEC : ebytes, 0x00020000);
  const int mrow = m0 + wm * WTM + fr;                          // this lane's pixel row for fragment i = 0
  const int ncol0 = n0 + wn * WTN + 8 * fq;                     // first of its 8 channels for pair a2 = 0
  const unsigned elem0 = (unsigned)mrow * (unsigned)a.Ncol + (unsigned)ncol0;   // element index of (mrow, ncol0)
  const unsigned rstep = 16u * (unsigned)a.Ncol;                // elements between fragment rows (scalar)
  // parity-class mode: fragment rows are class-local pixels; erow[i] = element index of (their real row, ncol0), or an
  // index whose byte offset lies past every tensor (loads give 0, stores are dropped) for rows past the class
  unsigned erow[FM];
  int mreal[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = mrow + 16 * i;
    if (s2c) {
      const int hw = Hc * Wc, n = m / hw, rem = m - n * hw, hq = rem / Wc, wq = rem - hq * Wc;
      mreal[i] = (n * a.Ho + 2 * hq + pa) * a.Wo + 2 * wq + pb;
      erow[i] = m < Mc ? (unsigned)mreal[i] * (unsigned)a.Ncol + (unsigned)ncol0 : (0x80000000u / (unsigned)OSZ);   // (x OSZ = 2 GiB: out of every descriptor)
      if (m >= Mc) mreal[i] = a.M;
    } else {
      mreal[i] = m;
      erow[i] = elem0 + (unsigned)i * rstep;
    }
  }
  unsigned addvo[ADD ? FM : 1];                                 // byte offset of (row i, ncol0) inside the addend
  unsigned addbytes = ybytes;
  if constexpr (ADD) {
    if (a.add_stride == 2) {
      const int Hc = (a.Ho + 1) >> 1, Wc = (a.Wo + 1) >> 1, hw = a.Ho * a.Wo;
      addbytes = (unsigned)a.N * Hc * Wc * a.Ncol * OSZ;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = mreal[i];
        const int n = m / hw, rem = m - n * hw, h = rem / a.Wo, w = rem - h * a.Wo;
        const bool on = m < a.M && !((h | w) & 1);
        addvo[i] = on ? (unsigned)((((n * Hc + (h >> 1)) * Wc + (w >> 1)) * a.Ncol + ncol0) * OSZ) : 0x80000000u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) addvo[i] = erow[i] * OSZ;
    }
  }
  const __amdgpu_buffer_rsrc_t rsrcAdd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ADD ? a.addend : a.Y), 0, addbytes, 0x00020000);
  constexpr bool STATS = (EPI == EPI_STATS || EPI == EPI_BNBWD || EPI == EPI_BNBWD_OUT);
  float csum[FN][4], csq[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { csum[j][r] = 0.f; csq[j][r] = 0.f; }
#pragma unroll
  for (int a2 = 0; a2 < FN / 2; ++a2) {
    const int nb = ncol0 + 32 * a2;
    float bias[8], emu[8], eis[8], esc[8], esh[8];
    if constexpr (EPI == EPI_FC) load8(a.bias + nb, bias);
    if constexpr (EPI == EPI_BNBWD || EPI == EPI_BNBWD_OUT) { load8(a.e_mean + nb, emu); load8(a.e_invstd + nb, eis); }
    if constexpr (EPI == EPI_BNBWD) { load8(a.e_scale + nb, esc); load8(a.e_shift + nb, esh); }
    // Issue every epilogue load of this channel pair first (up to 3 tensors x FM fragments in flight),
    // then consume: one memory round trip per pair instead of one per fragment.
    const unsigned eoff = elem0 + 32u * a2;                    // + i*rstep goes into the scalar offset
    // The loaded values stay PACKED (raw 16-byte registers, mask bits as one dword) until the row that uses them is
    // consumed: unpacking on arrival kept 3 x FM x 8 floats live at once and cost the conv1-type kernel its occupancy.
    constexpr bool EY = (EPI == EPI_BNBWD || EPI == EPI_BNBWD_OUT);
    Raw8<OUT32 ? 4 : 2> avr[ADD ? FM : 1];
    Raw8<sizeof(T)> yvr[EY ? FM : 1], ovr[EPI == EPI_BNBWD_OUT ? FM : 1];
    unsigned obits[EPI == EPI_BNBWD_OUT ? FM : 1];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      // (row i of the tile: a scalar offset from the lane's first row -- or, in parity-class mode, its own vector offset)
      const int so_t = s2c ? 0 : (int)(i * rstep * sizeof(T));
      const unsigned eoff_i = s2c ? erow[i] + 32u * a2 : eoff;
      if constexpr (ADD) avr[i].load(rsrcAdd, addvo[i] + 32u * a2 * OSZ, 0);
      if constexpr (EY) yvr[i].load(rsrcEy, eoff_i * (unsigned)sizeof(T), so_t);
      if constexpr (EPI == EPI_BNBWD_OUT) {
        if (ebits) {        // this lane's 8 channels are 8 / VEC groups: 1 mask byte (bf16) or 2 (fp32)
          const unsigned gidx = (s2c ? eoff_i : eoff + (unsigned)i * rstep) / (unsigned)VEC;
          if constexpr (VEC == 8) obits[i] = __builtin_amdgcn_raw_buffer_load_b8(rsrcEo, gidx, 0, 0);
          else obits[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsrcEo, gidx, 0, 0) |
                          ((unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsrcEo, gidx + 1u, 0, 0) << 4);
        } else {
          ovr[i].load(rsrcEo, eoff_i * (unsigned)sizeof(T), so_t);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const unsigned sto = s2c ? (erow[i] + 32u * a2) * OSZ : eoff * OSZ + (unsigned)(i * rstep * OSZ);   // byte offset of this row's store
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * a2 + (e >> 2)][e & 3];
      float yv_i[8];
      if constexpr (EY) yvr[i].unpack(yv_i);
      if constexpr (EPI == EPI_FC) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias[e];
      }
      if constexpr (ADD) {
        float av_i[8];
        avr[i].unpack(av_i);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += av_i[e];
      }
      if constexpr (EPI == EPI_BNBWD_OUT) {
        if (ebits) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((obits[i] >> e) & 1u) ? v[e] : 0.f;
        } else {
          float ov_i[8];
          ovr[i].unpack(ov_i);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ov_i[e] > 0.f ? v[e] : 0.f;
        }
      }
      if constexpr (EPI == EPI_BNBWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(yv_i[e], esc[e], esh[e]) > 0.f ? v[e] : 0.f;
      }
      if constexpr (OUT32) {
        u32x4_t s0, s1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] = __float_as_uint(v[e]); s1[e] = __float_as_uint(v[4 + e]); }
        // NOTE soffset must stay the literal 0 on stores: with an SGPR soffset hipcc (ROCm 7.2) assumes the
        // ">64-bit store data overwritten by the next VALU" hazard away and re-uses the data registers in
        // the very next instruction; on gfx950 that corrupts sporadic dwords once several blocks share a CU.
        __builtin_amdgcn_raw_buffer_store_b128(s0, rsrcY, sto, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(s1, rsrcY, sto + 16u, 0, 0);
      } else {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) { t[e] = (bf16_t)v[e]; v[e] = (float)t[e]; }   // stats of what the next layer reads
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<u32x4_t*>(&t), rsrcY, sto, 0, 0);
      }
      if constexpr (EPI == EPI_STATS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { csum[2 * a2 + (e >> 2)][e & 3] += v[e]; csq[2 * a2 + (e >> 2)][e & 3] += v[e] * v[e]; }
      }
      if constexpr (EPI == EPI_BNBWD || EPI == EPI_BNBWD_OUT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          csum[2 * a2 + (e >> 2)][e & 3] += v[e];
          csq[2 * a2 + (e >> 2)][e & 3] += v[e] * (yv_i[e] - emu[e]) * eis[e];
        }
      }
    }
  }
  if constexpr (STATS) {
    // Reduce over the 16 pixel-lanes with a halving butterfly: afterwards lane fr holds the total of
    // value index (fr % NV), NV = FN*4 values per lane.
    constexpr int NV = FN * 4;
    float vs[NV], vq[NV];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { vs[j * 4 + r] = csum[j][r]; vq[j * 4 + r] = csq[j][r]; }
    lane16_butterfly<NV, 8>(vs, vq, fr);
    // lane fr now owns value index vi = fr % NV  ->  fragment j = vi>>2, reg r = vi&3
    if (fr < NV) {
      const int j = fr >> 2, r = fr & 3;
      const int col = wn * WTN + 32 * (j >> 1) + 8 * fq + 4 * (j & 1) + r;
      red[(0 * WM + wm) * BN + col] = vs[0];
      red[(1 * WM + wm) * BN + col] = vq[0];
    }
    __syncthreads();
    static_assert(NT >= 2 * BN, "one thread per (statistic, column)");
    if (tid < 2 * BN) {
      const int which = tid / BN, col = tid % BN;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) sum += red[(which * WM + w) * BN + col];
      if (a.stat_tot) stat_run += sum;              // (flushed once per block, after its last tile)
      else a.stat_partial[((long)mt * 2 + which) * a.Ncol + n0 + col] = sum;
    }
    n0_blk = n0;
  }
  first_tile = false;
  };      // run_tile
  if constexpr (PERSIST) {
    // A tile loop makes every block-uniform value of a tile loop-invariant: hipcc hoists them ALL (the ~100 argument words,
    // 8 buffer descriptors, ...) and the 8-wave tiles, capped at 128 registers, then spill 100+ scalars and 20-30 vector
    // registers.  So each tile reads its arguments through a pointer to the kernarg segment (ConvArgs is the kernel's only
    // argument: offset 0) that an empty asm re-defines per tile: nothing derived from it can leave the loop body.
    typedef const __attribute__((address_space(4))) ConvArgs* kargp_t;
    for (int vb = blockIdx.x; vb < ka.nvb; vb += gridDim.x) {
      if (vb != (int)blockIdx.x) __syncthreads();       // the next tile re-uses the LDS stages and `red`
      kargp_t ap = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ap));
      run_tile(*ap, vb);
    }
  } else {
    // (one tile per block.  The loop above makes every block-uniform value of a tile loop-invariant: hipcc hoists them all
    // and the 8-wave tiles, capped at 128 registers, spill 100+ scalars and 20-30 vector registers -- so launches that
    // do not need the loop do not get it)
    run_tile(ka, blockIdx.x);
  }
  if constexpr (STATS_) {
    // (launch_igemm guarantees Ncol % BN == 0, so columns n0_blk .. n0_blk + BN - 1 exist; a padding block -- its virtual
    // tile lies past tilesM, `first_tile` still set -- has nothing to add)
    if (ka.stat_tot && tid_ < 2 * BN && !first_tile)      // replica = block index mod R (the XCD id for R = 8: as spread as the row-tile index)
      __hip_atomic_fetch_add(ka.stat_tot + ((long)(blockIdx.x % ka.stat_R) * 2 + tid_ / BN) * ka.Ncol + n0_blk + tid_ % BN, stat_run,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef FRX_DBG_TIMES
  __builtin_amdgcn_s_waitcnt(0);
  FRX_STAMP(3);
#endif
}

// ------------------------------------------------------------------------------------------
// wgrad.  Contraction runs over pixels, for which BOTH operands are strided in memory
// (channels are contiguous), so tiles are staged pixel-major [32 px][channels] and the MFMA
// fragments come out of LDS through the transposing read ds_read_b64_tr_b16 (bf16) or plain
// 4-byte reads (fp32 MFMA takes one float per lane).  grid = (co tiles * ci tiles, taps, splits).
// ------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* X;          // activations [N,Hx,Wx,Ci] (or padded NHWC4 for the stem)
  const void* dY;         // [M][Co]
  float* dW;              // fp32 [Co][R][S][Ci] (stem: [Co][R][8][4]); accumulated with atomics
  const float* in_scale;  // prologue on X, as in forward
  const float* in_shift;
  int in_relu;
  const void* dY2;        // YPRO: dY = alpha*dz + beta*y + gam  with dz = dY argument, y = dY2 (same indexing)
  const float* y_coef;    // YPRO: [3][Co] (alpha, beta, gam) from frx_bn_bwd_finalize
  int N, Hx, Wx, Ci, Ho, Wo, Co, R, S, stride, pad, M;
  int stem;
  int tilesCo, tilesCi;
  int chunks_per_split;   // K-chunks (of KP pixels) per split
  int splits;
  unsigned xbytes, ybytes;  // sizes of X and dY (dY2) in bytes: buffer-load bounds
  int variant;              // k_wgrad_grouped: WGV_* bits selecting the instantiation for this layer
  float* xsum;              // GRAM jobs (YM == 2): [Ci] column sums of the staged X operand, added with atomics
};

// XOR swizzle of the 32-byte slot inside a pixel row so the 8 pixel rows a half-wave touches
// in one transposing read fall on distinct banks (row bytes RB = 2*BT for bf16).
template <int RB> __device__ __forceinline__ int tr_swz(int row) {
  if constexpr (RB >= 256) return (((row & 3) | (((row >> 3) & 1) << 2)) << 5);
  else return ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 5);
}

enum { WG_POINTWISE = 0, WG_GENERAL = 1, WG_STEM = 2 };

// WMODE: gather geometry of the X operand (compile-time: keeps the K loop straight-line so the
// register ring gets counted vmcnt waits).  PRO: BN+ReLU prologue on X; a thread's channel group
// is the same for every chunk, so its 8 scale / 8 shift values stay in registers.
constexpr int WGRAD_SMEM = 2 * 2 * 32 * 256;       // bytes of LDS a 128 x 128 tile's two stages take (bf16 and fp32 alike)

// One (co tile, ci tile, tap, pixel split) work item; all four indices block-uniform (SGPRs).
// YM: the dY operand -- 0 as stored; 1 (YPRO) the BN backward alpha*dz + beta*y + gam of two tensors; 2 (GRAM) the X operand
// itself, prologue included: the tile then holds G = sum_m x[m,:]^T x[m,:] (Ci <= BT: one tile), and the block also adds the
// column sums of x into a.xsum.  What G and the sums are for: the weight gradient of a 1x1 conv whose dy is the BN backward
// of (dz, y) splits into  alpha * (dz^T x) + beta * (W G) + gam (x) sum(x)  because y = x W^T is linear in x -- the second
// full-width tensor y never has to be read (frx_wgrad_gram_finish closes it).
template <typename T, int BT, int WMODE, bool PRO, int YM>   // BT x BT output tile (co x ci)
__device__ __forceinline__ void wgrad_block(const WgradArgs& a, int split, int tile, int tap, char* smem) {
  constexpr int VEC = TT<T>::VEC;
  constexpr int KP = (sizeof(T) == 2) ? 32 : 16;   // pixels per K-chunk
  constexpr int RB = BT * sizeof(T);               // bytes per pixel row of a tile
  constexpr int CPR = RB / 16;                     // 16-byte chunks per row
  constexpr int LD = (KP * CPR) / 256;             // 16-byte loads per thread per operand
  constexpr int WT = BT / 2, F = WT / 16;          // 2x2 waves
  constexpr int PD = 3;                            // register ring depth (chunks in flight)
  constexpr bool YPRO = YM == 1, GRAM = YM == 2;
  static_assert(!GRAM || WMODE == WG_POINTWISE, "gram jobs: 1x1 / stride 1");
  static_assert(LD >= 1 && 256 % CPR == 0, "tile shape");
  static_assert(2 * 2 * KP * RB <= WGRAD_SMEM, "LDS budget");

  const int cot = __builtin_amdgcn_readfirstlane(tile / a.tilesCi), cit = tile - cot * a.tilesCi;
  const int tr_ = __builtin_amdgcn_readfirstlane(tap / a.S), ts_ = tap - tr_ * a.S;
  const int co0 = cot * BT, ci0 = cit * BT;
  const int nchunks = (a.M + KP - 1) / KP;
  const int kbeg = split * a.chunks_per_split;
  const int kend = min(nchunks, kbeg + a.chunks_per_split);
  if (kbeg >= kend) return;
  const int nk = kend - kbeg;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
  const T* __restrict__ dY = reinterpret_cast<const T*>(a.dY);
  const int hw = a.Ho * a.Wo;
  const int xch = (WMODE == WG_STEM) ? 32 : a.Ci;   // elements addressable in the X row for this tap
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)a.Wo;

  // fixed per thread: its 16-byte column and its first row inside a chunk (row step 256/CPR per load)
  const int ch = tid % CPR, row0 = tid / CPR;
  constexpr int RSTEP = 256 / CPR;
  const int co = co0 + ch * VEC, ci = ci0 + ch * VEC;
  const bool cook = co < a.Co, ciok = ci < xch;
  // per-channel constants as 16-byte loads (channel groups past the tensor read group 0: their columns are never stored)
  float psc[VEC], psh[VEC];
  if constexpr (PRO) {
    const int cs = ciok ? ci : 0;
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const float4 s4 = *reinterpret_cast<const float4*>(a.in_scale + cs + 4 * q), h4 = *reinterpret_cast<const float4*>(a.in_shift + cs + 4 * q);
      psc[4 * q] = s4.x; psc[4 * q + 1] = s4.y; psc[4 * q + 2] = s4.z; psc[4 * q + 3] = s4.w;
      psh[4 * q] = h4.x; psh[4 * q + 1] = h4.y; psh[4 * q + 2] = h4.z; psh[4 * q + 3] = h4.w;
    }
  }
  float yal[VEC], ybe[VEC], yga[VEC];          // YPRO: BN-backward coefficients of this thread's output channels
  if constexpr (YPRO) {
    const int cs = cook ? co : 0;
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const float4 a4 = *reinterpret_cast<const float4*>(a.y_coef + cs + 4 * q), b4 = *reinterpret_cast<const float4*>(a.y_coef + a.Co + cs + 4 * q),
                   g4 = *reinterpret_cast<const float4*>(a.y_coef + 2 * a.Co + cs + 4 * q);
      yal[4 * q] = a4.x; yal[4 * q + 1] = a4.y; yal[4 * q + 2] = a4.z; yal[4 * q + 3] = a4.w;
      ybe[4 * q] = b4.x; ybe[4 * q + 1] = b4.y; ybe[4 * q + 2] = b4.z; ybe[4 * q + 3] = b4.w;
      yga[4 * q] = g4.x; yga[4 * q + 1] = g4.y; yga[4 * q + 2] = g4.z; yga[4 * q + 3] = g4.w;
    }
  }
  // Buffer loads: per-lane byte offset fixed for the whole kernel (dY, pointwise X) plus a SCALAR offset that
  // walks the pixel axis; rows past M and columns past the tensor fall outside the descriptor and read 0.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dY), 0, a.ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcY2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(YPRO ? a.dY2 : a.dY), 0, a.ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, a.xbytes, 0x00020000);
  unsigned yvoff[LD], xvoff[LD];
  int gn[LD], goh[LD], gow[LD];                 // WG_GENERAL / WG_STEM: output pixel of each staged row, walked incrementally
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    const int r = row0 + RSTEP * i;
    yvoff[i] = cook ? (unsigned)((r * a.Co + co) * (int)sizeof(T)) : OOB;
    xvoff[i] = ciok ? (unsigned)((r * a.Ci + ci) * (int)sizeof(T)) : OOB;      // pointwise only
    if constexpr (WMODE != WG_POINTWISE) {
      const int m = kbeg * KP + r;
      gn[i] = m / hw;
      const int rem = m - gn[i] * hw;
      goh[i] = rem / a.Wo;
      gow[i] = rem - goh[i] * a.Wo;
    }
  }
  // one chunk = KP pixels further along (n, oh, ow): a mixed-radix add with single carries (branch-free -- a
  // `while` here put two divergent loops into every K-chunk of the 3x3 layers)
  const int step_n = KP / hw, step_h = (KP % hw) / a.Wo, step_w = (KP % hw) % a.Wo;

  uint4 ry[GRAM ? 1 : PD][LD], rx[PD][LD];
  uint4 ry2[YPRO ? PD : 1][LD];
  float xs_acc[GRAM ? VEC : 1];        // GRAM: this thread's running column sums (its 16-byte channel group, its rows)
#pragma unroll
  for (int e = 0; e < (GRAM ? VEC : 1); ++e) xs_acc[e] = 0.f;
  unsigned rmask[PD];     // WG_GENERAL / WG_STEM: bit i = the gathered X row of load i is inside the image

  auto issue_chunk = [&](int kc, auto slot_tag) {
    constexpr int slot = decltype(slot_tag)::value;
    unsigned msk = 0;
    const int soy = kc * KP * a.Co * (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      if constexpr (!GRAM) ry[slot][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcY, yvoff[i], soy, 0));
      if constexpr (YPRO) ry2[slot][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcY2, yvoff[i], soy, 0));
      if constexpr (WMODE == WG_POINTWISE) {
        rx[slot][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX, xvoff[i], kc * KP * a.Ci * (int)sizeof(T), 0));
      } else {
        bool xok = ciok && kc * KP + row0 + RSTEP * i < a.M;
        unsigned voff;
        if constexpr (WMODE == WG_STEM) {
          voff = (unsigned)((((gn[i] * a.Hx + goh[i] * 2 + tr_) * a.Wx + gow[i] * 2) * 4 + ci) * (int)sizeof(T));
        } else {
          const int hi = goh[i] * a.stride - a.pad + tr_, wi = gow[i] * a.stride - a.pad + ts_;
          xok = xok && (unsigned)hi < (unsigned)a.Hx && (unsigned)wi < (unsigned)a.Wx;
          voff = (unsigned)((((gn[i] * a.Hx + hi) * a.Wx + wi) * a.Ci + ci) * (int)sizeof(T));
        }
        rx[slot][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX, xok ? voff : OOB, 0, 0));
        // walk this row's pixel to the next chunk
        gow[i] += step_w;
        const int cw = gow[i] >= a.Wo ? 1 : 0;
        gow[i] -= cw ? a.Wo : 0;
        goh[i] += step_h + cw;
        const int chh = goh[i] >= a.Ho ? 1 : 0;
        goh[i] -= chh ? a.Ho : 0;
        gn[i] += step_n + chh;
        msk |= (xok ? 1u : 0u) << i;
      }
    }
    rmask[slot] = msk;
  };
  // Registers -> LDS.  Rows that are not real pixels load as 0, but a prologue turns 0 into f(0): a product
  // vanishes when EITHER factor is 0, so one operand is forced back to 0 there -- X wherever its gather can leave
  // the image (3x3 / stem: every chunk), and for pointwise layers only the chunk that straddles M (TAIL).
  auto commit_chunk = [&](int buf, auto slot_tag, int kc, auto tail_tag) {
    constexpr int slot = decltype(slot_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    char* Ys = smem + buf * (2 * KP * RB);
    char* Xs = Ys + KP * RB;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int row = row0 + RSTEP * i;
      const int off = row * RB + ((ch * 16) ^ (sizeof(T) == 2 ? tr_swz<RB>(row) : 0));
      uint4 vy = GRAM ? rx[slot][i] : ry[slot][i], vx = rx[slot][i];
      if constexpr (PRO) {
        vx = bn_relu_vec<T>(vx, psc, psh, a.in_relu);
        if constexpr (WMODE != WG_POINTWISE) {
          if (!((rmask[slot] >> i) & 1u)) vx = make_uint4(0, 0, 0, 0);
        } else if constexpr (TAIL && !YPRO) {
          if (kc * KP + row >= a.M) vx = make_uint4(0, 0, 0, 0);
        }
      }
      if constexpr (YPRO) {
        vy = affine2_vec<T>(vy, ry2[slot][i], yal, ybe, yga);
        if constexpr (TAIL) {
          if (kc * KP + row >= a.M) vy = make_uint4(0, 0, 0, 0);
        }
      }
      if constexpr (GRAM) {        // one staged tile serves as both operands; its column sums on the way
        if constexpr (TAIL && !PRO) { if (kc * KP + row >= a.M) vx = make_uint4(0, 0, 0, 0); }      // (rows past M load as 0 already; kept for symmetry)
        *reinterpret_cast<uint4*>(Ys + off) = vx;
        if constexpr (sizeof(T) == 2) {
          const unsigned* w = reinterpret_cast<const unsigned*>(&vx);
#pragma unroll
          for (int e = 0; e < 4; ++e) { xs_acc[2 * e] += __uint_as_float(w[e] << 16); xs_acc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        } else {
          const float* w = reinterpret_cast<const float*>(&vx);
#pragma unroll
          for (int e = 0; e < 4; ++e) xs_acc[e] += w[e];
        }
      } else {
        *reinterpret_cast<uint4*>(Ys + off) = vy;
        *reinterpret_cast<uint4*>(Xs + off) = vx;
      }
    }
  };

  f32x4 acc[F][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto compute_chunk = [&](int cur) {
    const char* Ys = smem + cur * (2 * KP * RB);
    const char* Xs = GRAM ? Ys : Ys + KP * RB;
    if constexpr (sizeof(T) == 2) {
      // lane (16g + 4q + p) supplies &tile[pix0 + q][col0 + 4p]; it receives column (lane&15),
      // rows pix0..pix0+3.  Two reads (pix0 = 8g, 8g+4) build the k = 8g..8g+7 fragment.
      const int q = (lane >> 2) & 3, p = lane & 3;
      s16x4 ya[F][2], xb[F][2];
#pragma unroll
      for (int i = 0; i < F; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = 8 * fq + 4 * h + q;
          const int colb = (wr * WT + i * 16 + 4 * p) * 2;
          ya[i][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(Ys + row * RB + (colb ^ tr_swz<RB>(row))));
          const int colx = (wc * WT + i * 16 + 4 * p) * 2;
          xb[i][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(Xs + row * RB + (colx ^ tr_swz<RB>(row))));
        }
#pragma unroll
      for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) {
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 av = __builtin_shufflevector(ya[i][0], ya[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
          s16x8 bv = __builtin_shufflevector(xb[j][0], xb[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&av),
                                                              *reinterpret_cast<bf16x8*>(&bv), acc[i][j], 0, 0, 0);
        }
    } else {
      const float* Yf = reinterpret_cast<const float*>(Ys);
      const float* Xf = reinterpret_cast<const float*>(Xs);
#pragma unroll
      for (int ks = 0; ks < KP / 4; ++ks) {
        float ya[F], xb[F];
        const int row = ks * 4 + fq;
#pragma unroll
        for (int i = 0; i < F; ++i) {
          ya[i] = Yf[row * BT + wr * WT + i * 16 + fr];
          xb[i] = Xf[row * BT + wc * WT + i * 16 + fr];
        }
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
          for (int j = 0; j < F; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[i], xb[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  static_assert(PD == 3, "ring written out for 3 slots");
  // chunk j (relative to kbeg) lives in ring slot j % 3 and LDS stage j & 1
  using STEADY = std::false_type;
  using TAILC = std::true_type;
  issue_chunk(kbeg, S0{});
  if (1 < nk) issue_chunk(kbeg + 1, S1{});
  if (2 < nk) issue_chunk(kbeg + 2, S2{});
  commit_chunk(0, S0{}, kbeg, TAILC{});
  __syncthreads();
  FRX_STAMP(1);
  int j = 0;
  for (; j + 5 < nk; j += 3) {                 // steady state: no branch between issue and wait
    // (sched_barrier: register-only work may otherwise be hoisted across s_barrier -- hipcc moved the unpacking of
    // the NEWEST ring slot to the top of the iteration, i.e. a vmcnt(0) that drains the ring every three chunks)
    compute_chunk(j & 1);
    issue_chunk(kbeg + j + 3, S0{});
    commit_chunk((j + 1) & 1, S1{}, kbeg + j + 1, STEADY{});
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    compute_chunk((j + 1) & 1);
    issue_chunk(kbeg + j + 4, S1{});
    commit_chunk((j + 2) & 1, S2{}, kbeg + j + 2, STEADY{});
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    compute_chunk((j + 2) & 1);
    issue_chunk(kbeg + j + 5, S2{});
    commit_chunk((j + 3) & 1, S0{}, kbeg + j + 3, STEADY{});
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; j < nk; j += 3) {
    {
      compute_chunk(j & 1);
      if (j + 3 < nk) issue_chunk(kbeg + j + 3, S0{});
      if (j + 1 < nk) commit_chunk((j + 1) & 1, S1{}, kbeg + j + 1, TAILC{});
      __syncthreads();
    }
    if (j + 1 < nk) {
      compute_chunk((j + 1) & 1);
      if (j + 4 < nk) issue_chunk(kbeg + j + 4, S1{});
      if (j + 2 < nk) commit_chunk((j + 2) & 1, S2{}, kbeg + j + 2, TAILC{});
      __syncthreads();
    }
    if (j + 2 < nk) {
      compute_chunk((j + 2) & 1);
      if (j + 5 < nk) issue_chunk(kbeg + j + 5, S2{});
      if (j + 3 < nk) commit_chunk((j + 3) & 1, S0{}, kbeg + j + 3, TAILC{});
      __syncthreads();
    }
  }

  FRX_STAMP(2);
  if constexpr (GRAM) {
    // column sums: the 256 / CPR threads that share a channel group meet in LDS (the stages are free: last barrier passed)
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[(row0 * CPR + ch) * VEC + e] = xs_acc[e];
    __syncthreads();
    if (tid < CPR * VEC && a.xsum) {
      float t = 0.f;
      for (int r = 0; r < RSTEP; ++r) t += red[r * CPR * VEC + tid];
      const int c = ci0 + tid;
      if (c < a.Ci) atomicAdd(a.xsum + c, t);
    }
    __syncthreads();
  }
  const int ldw = (WMODE == WG_STEM) ? 32 : a.Ci;   // elements per (co, r, s-row) line of dW
  const bool atomic = a.splits > 1;
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const int ci = ci0 + wc * WT + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wr * WT + i * 16 + fq * 4 + r;
        // stem: the 8th tap and the 4th channel of the [7][8][4] layout are padding pinned to zero -- leave
        // their gradient slots alone (they stay 0 from zero_grad) instead of zeroing them after the fact
        const bool live = (WMODE != WG_STEM) || ((ci >> 2) < 7 && (ci & 3) < 3);
        if (co < a.Co && ci < ldw && live) {
          const long o = (WMODE == WG_STEM) ? (((long)co * a.R + tr_) * 32 + ci)
                                : ((((long)co * a.R + tr_) * a.S + ts_) * a.Ci + ci);
          if (atomic) atomicAdd(a.dW + o, acc[i][j][r]); else a.dW[o] += acc[i][j][r];
        }
      }
    }
}

template <typename T, int BT, int WMODE, bool PRO, int YM>
__global__ __launch_bounds__(256, 2) void k_wgrad(WgradArgs a) {
  FRX_STAMP(0);
  __shared__ __attribute__((aligned(16))) char smem[WGRAD_SMEM];
  // 1-D grid, split index fastest: blocks that stream the SAME pixel range (same split, other
  // tiles / taps) are `splits` apart in dispatch order, i.e. on one XCD when splits % 8 == 0.
  // (integer division is expanded on the vector ALU: pin the block-uniform results back into SGPRs, or every
  // buffer load that takes one of them as its scalar offset is wrapped in a waterfall loop)
  const int split = __builtin_amdgcn_readfirstlane(blockIdx.x % a.splits);
  const int rest = __builtin_amdgcn_readfirstlane(blockIdx.x / a.splits);
  const int tile = __builtin_amdgcn_readfirstlane(rest % (a.tilesCo * a.tilesCi));
  const int tap = __builtin_amdgcn_readfirstlane(rest / (a.tilesCo * a.tilesCi));
  wgrad_block<T, BT, WMODE, PRO, YM>(a, split, tile, tap, smem);
#ifdef FRX_DBG_TIMES
  __builtin_amdgcn_s_waitcnt(0);
  FRX_STAMP(3);
#endif
}

// Every weight gradient of (a part of) the backward pass in ONE launch: persistent blocks walk a work list of
// (layer, tile, tap, pixel split) items.  The items' fire-and-forget atomics drain while the block already
// loads its next item, per-launch floors (53 of them) and the tails of 128-block launches disappear, and
// one list balances layers of very different size across the whole GPU.
struct WgradItem { int layer, split, tile, tap; };
enum { WGV_BT128 = 1, WGV_GENERAL = 2, WGV_STEM = 4, WGV_PRO = 8, WGV_YPRO = 16, WGV_GRAM = 32 };   // WgradArgs::variant bits

// SMALL: a list whose layers all take the 64 x 64 tile (layer1 and the stem: the HBM-bound end of the network) gets its own
// instantiation -- under 128 registers, four blocks per CU -- instead of the register budget of the widest 128 x 128 variant.
template <typename T, bool SMALL>
__global__ __launch_bounds__(256, SMALL ? 4 : 2) void k_wgrad_grouped(const WgradArgs* __restrict__ layers,
                                                                      const WgradItem* __restrict__ items, int nitems,
                                                                      int* __restrict__ counters) {
  __shared__ __attribute__((aligned(16))) char smem[WGRAD_SMEM];
  __shared__ int s_next;
  // Items are DRAWN, not strided: a block takes the next entry of its XCD's list (item 8 k + xcd, k from an atomic
  // counter per XCD) whenever it is free.  The tiles of one (layer, split) are adjacent entries, so they start within
  // one draw of each other whatever the earlier items cost and stream their shared pixel range through that XCD's L2
  // together; with a fixed stride the blocks drift apart over the rounds.  Every block ends on exactly one failed
  // draw, so the draw numbered (entries + blocks - 1) is the list's last and re-zeroes the counter for the next launch.
  const int xcd = blockIdx.x & 7, per_xcd = nitems >> 3, blocks_xcd = gridDim.x >> 3;
  for (int it = blockIdx.x;; it += gridDim.x) {
    if (counters) {
      if (threadIdx.x == 0) {
        const int k = __hip_atomic_fetch_add(counters + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == per_xcd + blocks_xcd - 1) __hip_atomic_store(counters + xcd, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_next = k;
      }
      __syncthreads();
      const int k = s_next;
      if (k >= per_xcd) break;
      it = 8 * k + xcd;
    } else if (it >= nitems) {
      break;
    }
    const WgradItem w = items[it];
    const int li = __builtin_amdgcn_readfirstlane(w.layer);
    if (li < 0) { __syncthreads(); continue; }      // padding of the XCD-interleaved list (barrier: s_next is rewritten next)
    const WgradArgs a = layers[li];
    const int split = __builtin_amdgcn_readfirstlane(w.split), tile = __builtin_amdgcn_readfirstlane(w.tile),
              tap = __builtin_amdgcn_readfirstlane(w.tap);
    const int v = __builtin_amdgcn_readfirstlane(a.variant);
#define FRX_WGV(BT_, WM_, PRO_, YP_) wgrad_block<T, BT_, WM_, PRO_, YP_>(a, split, tile, tap, smem)
    if (v & WGV_STEM) FRX_WGV(64, WG_STEM, false, 0);
    else if (v & WGV_GRAM) {       // (1x1 / stride 1, one tile: G = x^T x and the column sums of x)
      if (!SMALL && (v & WGV_BT128)) {
        if constexpr (SMALL) {}
        else if (v & WGV_PRO) FRX_WGV(128, WG_POINTWISE, true, 2); else FRX_WGV(128, WG_POINTWISE, false, 2);
      } else {
        if (v & WGV_PRO) FRX_WGV(64, WG_POINTWISE, true, 2); else FRX_WGV(64, WG_POINTWISE, false, 2);
      }
    }
    else if (!SMALL && (v & WGV_BT128)) {
      if constexpr (SMALL) {}
      else if ((v & WGV_GENERAL) && (v & WGV_YPRO)) { if (v & WGV_PRO) FRX_WGV(128, WG_GENERAL, true, true); else FRX_WGV(128, WG_GENERAL, false, true); }
      else if (v & WGV_GENERAL) { if (v & WGV_PRO) FRX_WGV(128, WG_GENERAL, true, false); else FRX_WGV(128, WG_GENERAL, false, false); }
      else if (v & WGV_YPRO) { if (v & WGV_PRO) FRX_WGV(128, WG_POINTWISE, true, true); else FRX_WGV(128, WG_POINTWISE, false, true); }
      else { if (v & WGV_PRO) FRX_WGV(128, WG_POINTWISE, true, false); else FRX_WGV(128, WG_POINTWISE, false, false); }
    } else {
      if ((v & WGV_GENERAL) && (v & WGV_YPRO)) { if (v & WGV_PRO) FRX_WGV(64, WG_GENERAL, true, true); else FRX_WGV(64, WG_GENERAL, false, true); }
      else if (v & WGV_GENERAL) { if (v & WGV_PRO) FRX_WGV(64, WG_GENERAL, true, false); else FRX_WGV(64, WG_GENERAL, false, false); }
      else if (v & WGV_YPRO) { if (v & WGV_PRO) FRX_WGV(64, WG_POINTWISE, true, true); else FRX_WGV(64, WG_POINTWISE, false, true); }
      else { if (v & WGV_PRO) FRX_WGV(64, WG_POINTWISE, true, false); else FRX_WGV(64, WG_POINTWISE, false, false); }
    }
#undef FRX_WGV
    __syncthreads();        // the next item re-uses the LDS stages
  }
}

}  // namespace frx
