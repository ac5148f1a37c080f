// Streamed pointwise kernels (bf16) for the "many channels in, few channels out" 1x1 launches of layer1 / layer2:
//   forward of a bottleneck's conv1 (256 -> 64, 256 -> 128, 512 -> 128: the block input is a stored bf16 tensor, no prologue,
//   BatchNorm statistics of the output into replicated totals), and
//   input gradient of its conv3 (256 <- 64 ... i.e. contraction over the 4x width: dy = alpha * dz + beta * y + gam of two
//   full-width tensors as the operand, the masked-statistics epilogue of bn2 behind it).
// Reference ops replaced: torchvision Bottleneck.conv1 forward / conv3 + bn3 backward (main_code/utils/backbones.py:16-18,
// run by model_utils.py:177,185).
//
// Such a launch is a stream over one or two full-width tensors with a small GEMM attached: every input element is used by
// ONE output pixel's row of N <= 128 outputs.  k_igemm stages it through LDS tile by tile (three K-chunks of prefetch per
// block, one barrier per chunk) and reaches 3.4-4.2 TB/s of the bytes.  Here nothing of the stream touches LDS: a wave owns
// 16 pixels at a time, its lanes load the MFMA operand fragments of those pixels straight from global memory (a lane's 16
// bytes are 8 consecutive channels of one pixel -- exactly the fragment), eight K-steps ahead and across the boundary to
// its next 16 pixels, so every wave keeps 8-16 KB in flight without a barrier anywhere in the loop; the whole weight
// matrix (32-128 KB) sits in LDS, read as fragments; the per-channel statistics stay in registers until the wave is done.
#include "conv_launch.h"

namespace frx {

template <int RB> __device__ __forceinline__ int pws_swz(int row) { return row & 15; }      // (rows of >= 512 bytes: see pw_rows.hip)

enum { PWS_FWD = 0, PWS_DGRAD = 1, PWS_MERGE = 2 };      // MERGE: the forward whose operand is the residual merge of the block before (frx_conv_fwd_merge)

template <int K, int N, int FLAVOUR>
__global__ __launch_bounds__(512, 2) void k_pw_stream(ConvArgs a, int units) {
  constexpr bool DGRAD = FLAVOUR == PWS_DGRAD, MERGE = FLAVOUR == PWS_MERGE, TWO = DGRAD || MERGE;      // TWO: two operand tensors
  typedef bf16_t T;
  constexpr int KS = K / 32, NF = N / 16, NA = N / 32;      // K-steps; 16-channel output fragments; 32-channel store groups
  constexpr int D = 8;                                        // K-steps a wave's loads run ahead
  constexpr int RB = K * 2;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(KS % D == 0 && N % 32 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                                    // [N][K] bf16, 16-byte slots XOR-swizzled by row
  float* sTab = reinterpret_cast<float*>(sW + N * RB);                // DGRAD: [K / 8][alpha, beta, gam][8]; MERGE: [K / 8][s3, sd, b3 + bd][8]
  float* sEpi = sTab + (TWO ? 3 * K : 0);                           // DGRAD: [N / 8][scale, shift][8] of the BN behind the output
  float* sStat = sEpi + (DGRAD ? 2 * N : 0);                          // [2][N]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(TWO ? a.X2 : a.X), 0, a.xbytes, 0x00020000);
  // MERGE: the block output and its > 0 bits (one byte per 16-byte channel group) leave as side outputs, each element once
  const __amdgpu_buffer_rsrc_t rsrcOut = __builtin_amdgcn_make_buffer_rsrc(MERGE ? a.dy_out : const_cast<void*>(a.X), 0, MERGE ? a.xbytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcMask = __builtin_amdgcn_make_buffer_rsrc((MERGE && a.mask_out) ? (void*)a.mask_out : const_cast<void*>(a.X), 0, (MERGE && a.mask_out) ? a.xbytes / 16u : 0u, 0x00020000);
  const unsigned ybytes = (unsigned)a.M * (unsigned)N * 2u;
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcEy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(DGRAD ? a.e_y : (const void*)a.Y), 0, ybytes, 0x00020000);

  FRX_STAMP(0);
  const int nwaves = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
  uint4 ra[D], ra2[TWO ? D : 1];
  // K-step ks of unit u (pixels 16 u .. 16 u + 15) into ring slot S
  auto issue = [&](int u, int ks, auto s_tag) {
    constexpr int S = decltype(s_tag)::value;
    const int p = u * 16 + fr;
    const unsigned off = p < a.M ? (unsigned)((p * K + ks * 32 + 8 * fq) * 2) : OOB;
    ra[S] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX, off, 0, 0));
    if constexpr (TWO) ra2[S] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX2, off, 0, 0));
  };
  // the first unit's first D steps are requested before anything else
  if (gw < units) {
    issue(gw, 0, std::integral_constant<int, 0>{}); issue(gw, 1, std::integral_constant<int, 1>{});
    issue(gw, 2, std::integral_constant<int, 2>{}); issue(gw, 3, std::integral_constant<int, 3>{});
    issue(gw, 4, std::integral_constant<int, 4>{}); issue(gw, 5, std::integral_constant<int, 5>{});
    issue(gw, 6, std::integral_constant<int, 6>{}); issue(gw, 7, std::integral_constant<int, 7>{});
  }

  // ---- set-up: weights -> LDS, tables
  {   // (N * K / 8 sixteen-byte pieces over 512 threads: 8 loads in flight per thread and round -- one at a time, the loop
      // cost a memory round trip per piece: 5-7 us of set-up)
    constexpr int PIECES = N * (K / 8), U = PIECES / 512 < 8 ? PIECES / 512 : 8, ROUNDS = PIECES / (512 * U);
    static_assert(PIECES % (512 * U) == 0, "weight pieces");
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = (r * U + u) * 512 + tid;
        v[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.W) + (size_t)q * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = (r * U + u) * 512 + tid;
        const int n = q / (K / 8), sl = q - n * (K / 8);
        // LDS row = position in FRAGMENT order (row 16 j + fr holds channel chan_of(j, fr)): the 16 lanes of a fragment read 16
        // consecutive rows, which the slot swizzle spreads over the banks (in channel order, rows r and r + 16 of a fragment collide)
        const int rr = 16 * (2 * (n >> 5) + ((n >> 2) & 1)) + 4 * ((n >> 3) & 3) + (n & 3);
        *reinterpret_cast<uint4*>(sW + rr * RB + ((sl ^ pws_swz<RB>(rr)) << 4)) = v[u];
      }
    }
  }
  for (int c = tid; c < 2 * N; c += 512) sStat[c] = 0.f;
  if constexpr (DGRAD) {
    if (a.in_scale) {
      for (int c = tid; c < K; c += 512) {
        float* t = sTab + (c >> 3) * 24 + (c & 7);
        t[0] = a.in_scale[c]; t[8] = a.in_shift[c]; t[16] = a.pro_gam[c];
      }
    } else {
      const BnTot b = bn_tot_copy(a.pro_tot);
      bn_tot_foreach<512>(b.tot, b.R, K, [&](int c, double sa, double sb) {
        float al, be, ga;
        bn_bwd_consts(sa, sb, b.inv_count, b.gamma[c], b.mean[c], b.invstd[c], al, be, ga);
        float* t = sTab + (c >> 3) * 24 + (c & 7);
        t[0] = al; t[8] = be; t[16] = ga;
      });
    }
    for (int c = tid; c < N; c += 512) {
      float* t = sEpi + (c >> 3) * 16 + (c & 7);
      t[0] = a.e_scale[c]; t[8] = a.e_shift[c];
    }
  }
  if constexpr (MERGE) {      // bn3's constants (and the projection BatchNorm's, if any) -> s3, sd, b3 + bd (as k_igemm's merge prologue)
    if (a.in_scale) {
      for (int c = tid; c < K; c += 512) {
        float* t = sTab + (c >> 3) * 24 + (c & 7);
        t[0] = a.in_scale[c]; t[8] = a.id_scale ? a.id_scale[c] : 1.f; t[16] = a.in_shift[c] + (a.id_shift ? a.id_shift[c] : 0.f);
      }
    } else {
      const BnTot b = bn_tot_copy(a.in_tot);
      bn_tot_foreach<512>(b.tot, b.R, K, [&](int c, double sm, double sq) {
        float mean, invstd, sc, sh; double var;
        bn_fwd_consts(sm, sq, b.inv_count, b.gamma[c], b.beta[c], b.eps, mean, invstd, sc, sh, var);
        float* t = sTab + (c >> 3) * 24 + (c & 7);
        t[0] = sc; t[8] = 1.f; t[16] = sh + 0.f;
      });
      if (a.id_tot.tot) {      // (the same thread owns channel c in both passes: no barrier in between)
        const BnTot bd = bn_tot_copy(a.id_tot);
        bn_tot_foreach<512>(bd.tot, bd.R, K, [&](int c, double sm, double sq) {
          float mean, invstd, sc, sh; double var;
          bn_fwd_consts(sm, sq, bd.inv_count, bd.gamma[c], bd.beta[c], bd.eps, mean, invstd, sc, sh, var);
          float* t = sTab + (c >> 3) * 24 + (c & 7);
          t[8] = sc; t[16] = t[16] + sh;
        });
      }
    }
  }
  __syncthreads();
  FRX_STAMP(1);

  float csum[NA][8], csq[NA][8];
#pragma unroll
  for (int g = 0; g < NA; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[g][e] = 0.f; csq[g][e] = 0.f; }

  for (int u = gw; u < units; u += nwaves) {
    const int p = u * 16 + fr;
    const bool pok = p < a.M;
    const int un = u + nwaves;                  // this wave's next unit
    const bool more = un < units;
    // the epilogue's operand (raw output of the BN behind this gradient) is requested with the unit, not after its K loop
    uint4 ey[DGRAD ? NA : 1];
    if constexpr (DGRAD) {
#pragma unroll
      for (int g = 0; g < NA; ++g)
        ey[g] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcEy, pok ? (unsigned)((p * N + 32 * g + 8 * fq) * 2) : OOB, 0, 0));
    }
    f32x4 acc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned char* sMaskW = reinterpret_cast<unsigned char*>(sStat + 2 * N) + wave * (16 * (K / 8));      // (MERGE) this wave's patch
    auto step = [&](int ks, auto s_tag) {
      constexpr int S = decltype(s_tag)::value;
      uint4 op = ra[S];
      if constexpr (DGRAD) {
        const float* t = sTab + (ks * 4 + fq) * 24;
        op = affine2_vec<T>(ra[S], ra2[S], t, t + 8, t + 16);
        if (!pok) op = make_uint4(0, 0, 0, 0);            // (pixels past M load as 0, which the affine map turns into gam)
      }
      if constexpr (MERGE) {
        const float* t = sTab + (ks * 4 + fq) * 24;
        unsigned mbits;
        op = merge_vec<T>(ra[S], ra2[S], t, t + 8, t + 16, mbits);
        if (!pok) op = make_uint4(0, 0, 0, 0);
        const unsigned so_b = pok ? (unsigned)((p * K + ks * 32 + 8 * fq) * 2) : OOB;
        u32x4_t sv; sv[0] = op.x; sv[1] = op.y; sv[2] = op.z; sv[3] = op.w;
        __builtin_amdgcn_raw_buffer_store_b128(sv, rsrcOut, so_b, 0, 0);
        // mask byte of (pixel, channel group 4 ks + fq): parked in the wave's LDS patch [16 pixels][K / 8 bytes]; stored as whole
        // 8-byte pieces after the unit (a byte store per step was 16 scattered 4-byte pieces per wave instruction)
        sMaskW[fr * (K / 8) + ks * 4 + fq] = (unsigned char)mbits;
      }
      // refill the slot: D steps on, in this unit or at the start of the next
      if (ks + D < KS) issue(u, ks + D, s_tag);
      else if (more) issue(un, ks + D - KS, s_tag);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int row = 16 * j + fr;
        const uint4 wf = *reinterpret_cast<const uint4*>(sW + row * RB + (((ks * 4 + fq) ^ pws_swz<RB>(row)) << 4));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf), *reinterpret_cast<bf16x8*>(&op), acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);          // (without it hipcc hoists every step's weight fragments to the top: 500+ registers)
    };
#pragma unroll 1
    for (int kb = 0; kb < KS; kb += D) {
      step(kb + 0, std::integral_constant<int, 0>{}); step(kb + 1, std::integral_constant<int, 1>{});
      step(kb + 2, std::integral_constant<int, 2>{}); step(kb + 3, std::integral_constant<int, 3>{});
      step(kb + 4, std::integral_constant<int, 4>{}); step(kb + 5, std::integral_constant<int, 5>{});
      step(kb + 6, std::integral_constant<int, 6>{}); step(kb + 7, std::integral_constant<int, 7>{});
    }
    if constexpr (MERGE) {      // the unit's mask bytes: pixel fr's K / 8 bytes lie contiguous; lane (fr, fq) stores piece fq of them
      static_assert(K / 8 == 32, "mask patch layout: four 8-byte pieces per pixel");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (wave-private: the wave's own writes, program order)
      const unsigned long long mv = *reinterpret_cast<const unsigned long long*>(sMaskW + fr * 32 + fq * 8);
      typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
      u32x2_t m2; m2[0] = (unsigned)mv; m2[1] = (unsigned)(mv >> 32);
      __builtin_amdgcn_raw_buffer_store_b64(m2, rsrcMask, pok ? (unsigned)(p * 32 + fq * 8) : OOB, 0, 0);
    }
    // ---- epilogue: fragments (2g, 2g + 1) hold this lane's channels 32 g + 8 fq + [0, 8) of pixel p
#pragma unroll
    for (int g = 0; g < NA; ++g) {
      float v[8], yv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[2 * g + (e >> 2)][e & 3];
      if constexpr (DGRAD) {
        const unsigned* q = reinterpret_cast<const unsigned*>(&ey[g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { yv[2 * e] = __uint_as_float(q[e] << 16); yv[2 * e + 1] = __uint_as_float(q[e] & 0xffff0000u); }
        const float* t = sEpi + (4 * g + fq) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(yv[e], t[e], t[8 + e]) > 0.f ? v[e] : 0.f;
      }
      bf16x8 tb;
#pragma unroll
      for (int e = 0; e < 8; ++e) tb[e] = (bf16_t)v[e];
      const u32x4_t tw = *reinterpret_cast<u32x4_t*>(&tb);
      __builtin_amdgcn_raw_buffer_store_b128(tw, rsrcY, pok ? (unsigned)((p * N + 32 * g + 8 * fq) * 2) : OOB, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {                              // statistics of what the next kernel reads
        const float vl = __uint_as_float(tw[e] << 16), vh = __uint_as_float(tw[e] & 0xffff0000u);
        csum[g][2 * e] += vl; csum[g][2 * e + 1] += vh;
        if constexpr (DGRAD) { csq[g][2 * e] += vl * yv[2 * e]; csq[g][2 * e + 1] += vh * yv[2 * e + 1]; }
        else { csq[g][2 * e] += vl * vl; csq[g][2 * e + 1] += vh * vh; }
      }
    }
  }
  FRX_STAMP(2);
  // ---- the wave's sums -> the block's (LDS) -> the replicated totals
#pragma unroll
  for (int g = 0; g < NA; ++g) {
    lane16_butterfly<8, 8>(csum[g], csq[g], fr);
    if (fr < 8) {
      const int col = 32 * g + 8 * fq + fr;
      atomicAdd(&sStat[col], csum[g][0]);
      atomicAdd(&sStat[N + col], csq[g][0]);
    }
  }
  __syncthreads();
  const int rep = blockIdx.x & (a.stat_R - 1);
  for (int c = tid; c < N; c += 512) {
    const float s1 = sStat[c], s2 = sStat[N + c];
    __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 0) * N + c, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 1) * N + c, DGRAD ? a.e_invstd[c] * (s2 - a.e_mean[c] * s1) : s2, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef FRX_DBG_TIMES
  __builtin_amdgcn_s_waitcnt(0);
  FRX_STAMP(3);
#endif
}

// (K, N) served.  Measured inside a training step against k_igemm: (256, 64) -- layer1 -- input gradient 54 vs 61 us, forward
// 33-36 vs 38 us.  The 128-output shapes of layer2 ((512, 128), (256, 128)) LOSE (forward 30 vs 22, input gradient 43 vs 36 us):
// with 1.5 units per wave a launch is set-up (weights into LDS behind the first operand loads: 7-10 us) plus flush, not stream;
// FRX_PW_STREAM=2 runs them anyway (tests, measurements).
static bool shape_ok(int K, int N, bool dgrad) {
  if (K == 256 && N == 64) return true;
  const char* e = getenv("FRX_PW_STREAM");
  return e && atoi(e) == 2 && ((K == 512 && N == 128) || (!dgrad && K == 256 && N == 128));
}

bool pw_stream_ok(const ConvArgs& a, int dtype, int epi) {
  if (const char* e = getenv("FRX_PW_STREAM")) { if (atoi(e) == 0) return false; }
  const bool pw = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && !a.s2c;
  if (dtype != FRX_BF16 || !pw || !a.stat_tot || a.stat_partial || a.out_f32 || a.bias || a.addend) return false;
  if (a.mode == MODE_FWD && a.X2)      // the merge prologue (frx_conv_fwd_merge): layer1's shape
    return a.dy_out && epi == EPI_STATS && a.Kc == 256 && (a.Ncol == 64 || a.Ncol == 128);      // layer1's conv1 and layer2's first conv1
  if (a.dy_out) return false;
  if (a.mode == MODE_FWD) return !a.in_scale && !a.in_tot.tot && epi == EPI_STATS && shape_ok(a.Kc, a.Ncol, false);
  if (a.mode == MODE_DGRAD) return a.X2 && epi == EPI_BNBWD && a.e_scale && a.e_shift && !a.e_out && !a.e_bits && shape_ok(a.Kc, a.Ncol, true);
  return false;
}

template <int K, int N, int FLAVOUR>
static void launch_one(hipStream_t st, const ConvArgs& a) {
  constexpr bool DGRAD = FLAVOUR == PWS_DGRAD, TWO = FLAVOUR != PWS_FWD;
  static bool attr_done[64] = {false};      // (per device: the attribute belongs to the function ON a device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw_stream<K, N, FLAVOUR>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    attr_done[dev & 63] = true;
  }
  const unsigned lds = (unsigned)(N * K * 2 + (TWO ? 3 * K * 4 : 0) + (DGRAD ? 2 * N * 4 : 0) + 2 * N * 4 + (FLAVOUR == PWS_MERGE ? 8 * 16 * (K / 8) : 0));
  const int units = cdiv(a.M, 16);
  const int blocks_per_cu = 1;      // (two blocks per CU where they fit were slower: the set-up is per block)
  int grid = 256 * blocks_per_cu;
  if (grid * 8 > units) grid = cdiv(units, 8);
  hipLaunchKernelGGL((k_pw_stream<K, N, FLAVOUR>), dim3(grid), dim3(512), lds, st, a, units);
}

int launch_pw_stream(hipStream_t st, const ConvArgs& a) {
  const bool dg = a.mode == MODE_DGRAD, mg = a.mode == MODE_FWD && a.X2;
  note_igemm_launch(16, a.Ncol, 8, 64, 0, a.mode, dg ? 2 : (mg ? 3 : 0), dg ? EPI_BNBWD : EPI_STATS, 0, 1, 3);
  if (mg) { if (a.Ncol == 64) launch_one<256, 64, PWS_MERGE>(st, a); else launch_one<256, 128, PWS_MERGE>(st, a); }
  else if (a.Kc == 256 && a.Ncol == 64) { if (dg) launch_one<256, 64, PWS_DGRAD>(st, a); else launch_one<256, 64, PWS_FWD>(st, a); }
  else if (a.Kc == 512 && a.Ncol == 128) { if (dg) launch_one<512, 128, PWS_DGRAD>(st, a); else launch_one<512, 128, PWS_FWD>(st, a); }
  else launch_one<256, 128, PWS_FWD>(st, a);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

}  // namespace frx

FRX_DBG_EXPORT(frx_debug_times_pw_stream)
