// C-ABI entry points for the implicit-GEMM convolutions (kernels in conv_kernels.h).
// Reference call sites replaced: every nn.Conv2d / nn.Linear of the torchvision ResNet-50 the
// reference builds in main_code/utils/backbones.py:16-18, forward and autograd backward
// (main_code/utils/criterion.py:320, model_utils.py:177,185).
#include "conv_launch.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace frx {

// Patch mode (conv_kernels.h "P3") serves the bf16 3x3 / stride 1 / pad 1 layers: images up to 30 pixels wide (the patch of a
// 128-pixel tile is 128 + 2 W + 2 rows of the 191 its buffer holds) and an even number of 64-byte channel chunks of the
// gathered operand.  FRX_CONV3X3=0: off (read per launch).
static bool p3_geometry(int dtype, int R, int S, int stride, int pad, int Wx, int Kc) {
  const char* e = getenv("FRX_CONV3X3");
  return !(e && atoi(e) == 0) && dtype == FRX_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && Wx <= 30 && Kc % 64 == 0;
}

// its tile: 128 pixels x (128 | 64) columns; 64 x 128 where 128-pixel tiles would leave CUs without one (pick_tile's threshold)
static void p3_tile(long M, int Ncol, int* bm, int* bn) {
  *bn = Ncol % 128 == 0 ? 128 : 64;
  *bm = (*bn == 128 && (long)cdiv(M, 128) * (Ncol / 128) < 192) ? 64 : 128;
}

static thread_local int t_last_launch[12] = {0};
void note_igemm_launch(int bm, int bn, int waves, int kc, int ns, int mode, int pro, int epi, int add, int persist, int spec) {
  const int v[12] = {bm, bn, waves, kc, ns, mode, pro, epi, add, persist, spec, t_last_launch[11]};
  memcpy(t_last_launch, v, sizeof(v));
}

// The fc layer (nn.Linear(2048, 512) on the pooled features, backbones.py:17; bf16 speed mode): M = batch rows are too few for
// the tiled kernel -- 32 tiles of 64 x 64, each a chain of 64 K-chunks: 21.6 us for 0.54 GFLOP (round 3).  Here one block owns
// ONE 16 x 16 output tile and its four waves split the contraction four ways; the MFMA fragments come straight from global
// memory (a lane's 16 bytes are 8 consecutive k of one row: no staging), 8 chunks in flight per wave; the four partial tiles
// meet in LDS in a fixed order (bit-reproducible).  512 blocks for 256 x 512 outputs.
__global__ __launch_bounds__(256) void k_fc_fwd_bf16(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ y, int M, int N, int K) {
  __shared__ float red[4][16][17];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  const int fr = lane & 15, fq = lane >> 4;
  const int row = m0 + fr < M ? m0 + fr : M - 1, col = n0 + fr < N ? n0 + fr : N - 1;      // (clamped: their outputs are not stored)
  const int kq = K / 4, kbeg = wave * kq;            // this wave's quarter of the contraction (host: K % 128 == 0)
  const bf16_t* xp = x + (long)row * K + kbeg + 8 * fq;
  const bf16_t* wp = w + (long)col * K + kbeg + 8 * fq;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kq; k0 += 256) {
    bf16x8 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 32 * u < kq ? k0 + 32 * u : 0;
      a[u] = *reinterpret_cast<const bf16x8*>(xp + k);
      b[u] = *reinterpret_cast<const bf16x8*>(wp + k);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + 32 * u < kq) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u], a[u], acc, 0, 0, 0);      // D[row = n][col = m]
  }
  // C/D map: col = lane & 15 (here: m), row = (lane >> 4) * 4 + reg (here: n)
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][fr][fq * 4 + r] = acc[r];
  __syncthreads();
  const int t = threadIdx.x, ml = t >> 4, nl = t & 15;
  if (m0 + ml < M && n0 + nl < N)
    y[(long)(m0 + ml) * N + n0 + nl] = (((red[0][ml][nl] + red[1][ml][nl]) + red[2][ml][nl]) + red[3][ml][nl]) + bias[n0 + nl];
}

static int launch_igemm(hipStream_t st, ConvArgs a, int dtype) {
  t_last_launch[11] = a.R * a.S * a.Kc;                 // contraction length (note_igemm_launch fills in the rest)
  FRX_CHECK_ARG(a.Ncol % 64 == 0, "igemm: output channel count %d must be a multiple of 64", a.Ncol);
  const size_t esz = dtype == FRX_BF16 ? 2 : 4;
  const size_t xb = (size_t)a.N * a.Hx * a.Wx * (a.mode == MODE_STEM ? 4 : a.Kc) * esz;
  const size_t wb = (size_t)a.Ncol * (a.mode == MODE_STEM ? a.R * 32 : a.R * a.S * a.Kc) * esz;
  const size_t yb = (size_t)a.M * a.Ncol * 4;
  FRX_CHECK_ARG(xb < 0x80000000ull && wb < 0x80000000ull && yb < 0x80000000ull,
                "igemm: tensors must stay below 2 GiB (32-bit buffer offsets)");
  a.xbytes = (unsigned)xb; a.wbytes = (unsigned)wb;
  const bool pointwise = a.mode != MODE_STEM && a.R == 1 && a.S == 1 && a.stride == 1;
  TileCfg c = pick_tile(a.M, a.Ncol, (long)a.R * a.S * a.Kc, pointwise, a.mode != MODE_STEM && !a.s2c);
  if (a.Ncol % c.bn != 0) c = TileCfg{c.bm, 64, 4, 64};
  FRX_CHECK_ARG(c.kc == 64 || (a.Kc * (int)esz) % c.kc == 0, "igemm: %d channels do not fill %d-byte K-chunks", a.Kc, c.kc);
  a.tilesM = cdiv(a.M, c.bm);
  if (a.s2c) {                     // tiles per parity class (h & 1, w & 1), class-major
    int t = 0;
    for (int cls = 0; cls < 4; ++cls) {
      a.cls_tile0[cls] = t;
      const long mc = (long)a.N * ((a.Ho - (cls >> 1) + 1) / 2) * ((a.Wo - (cls & 1) + 1) / 2);
      t += cdiv(mc, c.bm);
    }
    a.cls_tile0[4] = t;
    a.tilesM = t;
  }
  a.tilesN = cdiv(a.Ncol, c.bn);
  const bool has_pro = a.in_scale || a.in_tot.tot || a.X2;
  c.ns = dma_stages(c, !has_pro, a.mode == MODE_STEM, (long)a.tilesM * a.tilesN);
  a.nvb = (int)round_up(a.tilesM, 8) * a.tilesN;
  // Persistent blocks (conv_kernels.h: run_tile) when the launch has more tiles than fit on the chip at once and its
  // statistics (if any) go to replicated totals: `cap` resident blocks walk the tiles with a fixed stride.  cap / 8 is a
  // multiple of tilesN (a block stays in one column of tiles).
  int grid = a.nvb;
  {
    const int per_cu = c.waves == 8 ? 2 : 3;
    const int cap = per_cu * 256;
    // (the instantiated persistent variants: forward only.  Measured per launch inside a training step, persistent vs one
    // tile per block: the prologue-fed pointwise forwards of layer1/2 gain 7-9 % (the 64->256 conv3: 42.1 -> 38.9 us),
    // the input gradients LOSE 8-15 % -- their epilogues are where the registers run out, and a tile loop without
    // cross-tile prefetch only adds live state there: profiles/r03_persist_layer_times.txt)
    const bool tile_ok = c.kc == 64 && c.bm == 128 && !c.ns && a.mode == MODE_FWD && !a.X2;      // (the merge prologue has no persistent instantiation: it spills there)
    if (per_cu > 0 && tile_ok && !a.stat_partial && a.nvb > cap && (cap / 8) % a.tilesN == 0) grid = cap;
  }
  int epi = EPI_PLAIN;
  if (a.epi_bnbwd) epi = (a.e_out || a.e_bits) ? EPI_BNBWD_OUT : EPI_BNBWD;
  else if (a.stat_partial || a.stat_tot) epi = EPI_STATS;
  else if (a.out_f32 || a.bias) epi = EPI_FC;
  // Patch mode: launches with a prologue (forward: BN+ReLU, statistics or plain epilogue; input gradient: BN backward with
  // the masked-statistics epilogue), no addend.
  {
    bool p3 = a.mode != MODE_STEM && !a.s2c && p3_geometry(dtype, a.R, a.S, a.stride, a.pad, a.Wx, a.Kc) && a.Hx == a.Ho && a.Wx == a.Wo &&
              !a.addend && !a.out_f32 && !a.bias;
    // row tile: 128 pixels; 64 where that would leave CUs without a tile and the columns allow it (the same threshold as
    // pick_tile's).  Per-tile partial statistics are laid out by frx_conv_stat_rows, i.e. by pick_tile's row tile.
    int bm3, bn3;
    p3_tile(a.M, a.Ncol, &bm3, &bn3);
    p3 = p3 && (!a.stat_partial || c.bm == bm3);
    // (prologue-free input gradient -- dy materialised by a pass of its own -- on the 64-column tile only: layer1, where
    // the chunk-per-tap LDS-DMA kernel takes 43 us and the fused BN-backward prologue, two blocks per CU, 72)
    p3 = p3 && ((a.mode == MODE_FWD && has_pro && (epi == EPI_STATS || epi == EPI_PLAIN)) ||
                (a.mode == MODE_DGRAD && epi == EPI_BNBWD && (a.X2 || (!has_pro && bn3 == 64 && bm3 == 128))));
    if (p3) {
      a.tilesM = cdiv(a.M, bm3);
      a.tilesN = a.Ncol / bn3;
      a.nvb = (int)round_up(a.tilesM, 8) * a.tilesN;
      return launch_igemm_p3(st, a, epi, bm3, bn3);
    }
    FRX_CHECK_ARG(!(a.dy_out && a.mode == MODE_FWD && !a.X2),
                  "conv_fwd_keep: x_norm_out needs the patch-mode launch (partial-statistics rows only where frx_conv_tile's row tile is "
                  "frx_conv_patch_mode's)");
    FRX_CHECK_ARG(!(a.dy_out && (a.R != 1 || a.S != 1)),
                  "conv_dgrad_bn: pro_dy_out on a 3x3 needs the patch-mode launch (BN-backward prologue, masked-statistics epilogue, no "
                  "addend; partial-statistics rows only where frx_conv_tile's row tile is frx_conv_patch_mode's)");
  }
  if (epi == EPI_FC && dtype == FRX_BF16 && pointwise && a.mode == MODE_FWD && a.bias && a.out_f32 && !has_pro && a.M <= 4096 &&
      a.Kc % 128 == 0 && a.Kc * 2 % 16 == 0) {
    note_igemm_launch(16, 16, 4, 64, 0, MODE_FWD, 0, EPI_FC, 0, 0, 0);
    hipLaunchKernelGGL(k_fc_fwd_bf16, dim3(cdiv(a.Ncol, 16), cdiv(a.M, 16)), dim3(256), 0, st, (const bf16_t*)a.X, (const bf16_t*)a.W, a.bias,
                       (float*)a.Y, a.M, a.Ncol, a.Kc);
    FRX_LAUNCH_CHECK();
    return FRX_OK;
  }
  if (a.mode == MODE_STEM) return launch_igemm_stem(st, a, dtype, c, grid, epi);
  if (pw_rows_dgrad_ok(a, dtype, epi)) return launch_pw_rows_dgrad(st, a);
  if (pw_rows_fwd_ok(a, dtype, epi)) return launch_pw_rows_fwd(st, a);
  if (pw_stream_ok(a, dtype, epi)) return launch_pw_stream(st, a);
  if (a.mode == MODE_DGRAD)
    return a.X2 ? launch_igemm_dgrad_bn(st, a, dtype, c, grid, epi, a.addend != nullptr)
                : launch_igemm_dgrad_plain(st, a, dtype, c, grid, epi, a.addend != nullptr);
  FRX_CHECK_ARG(a.addend == nullptr, "igemm fwd: addend is a dgrad feature");
  if (a.X2) {      // the merge prologue (frx_conv_fwd_merge)
    FRX_CHECK_ARG(pointwise, "conv_fwd_merge: a 1x1 / stride 1 convolution");
    return launch_igemm_fwd(st, a, dtype, c, grid, 3, epi);
  }
  return launch_igemm_fwd(st, a, dtype, c, grid, (a.in_scale || a.in_tot.tot) ? 1 : 0, epi);
}

static int check_conv(const frx_conv_desc* d) {
  FRX_CHECK_ARG(d != nullptr, "conv desc is NULL");
  FRX_CHECK_ARG(d->dtype == FRX_F32 || d->dtype == FRX_BF16, "conv dtype %d unsupported", d->dtype);
  FRX_CHECK_ARG(d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->Ci > 0 && d->Co > 0, "conv dims must be positive");
  FRX_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv stride %d unsupported (1 or 2)", d->stride);
  const int ce = d->dtype == FRX_BF16 ? 32 : 16;
  if (d->stem) {
    FRX_CHECK_ARG(d->R == 7 && d->S == 7 && d->stride == 2 && d->pad == 3 && d->Ci == 3,
                  "stem conv must be 7x7 stride 2 pad 3 on 3 channels");
    FRX_CHECK_ARG(d->Ho == (d->Hi + 6 - 7) / 2 + 1 && d->Wo == (d->Wi + 6 - 7) / 2 + 1, "stem output size mismatch");
  } else {
    FRX_CHECK_ARG(d->Ci % ce == 0, "conv Ci=%d must be a multiple of %d for this dtype", d->Ci, ce);
    FRX_CHECK_ARG(d->Co % ce == 0, "conv Co=%d must be a multiple of %d for this dtype (dgrad contraction)", d->Co, ce);
    FRX_CHECK_ARG(d->Ho == (d->Hi + 2 * d->pad - d->R) / d->stride + 1 && d->Wo == (d->Wi + 2 * d->pad - d->S) / d->stride + 1,
                  "conv output size mismatch: got %dx%d", d->Ho, d->Wo);
  }
  FRX_CHECK_ARG((long)d->N * d->Hi * d->Wi < (1L << 31) / 4, "conv too large for 32-bit pixel indices");
  return FRX_OK;
}

}  // namespace frx
using namespace frx;

extern "C" int frx_conv_stat_rows(const frx_conv_desc* d) {
  if (check_conv(d) != FRX_OK) return -1;
  const long M = (long)d->N * d->Ho * d->Wo;
  return cdiv(M, pick_tile(M, d->Co, (long)d->R * d->S * d->Ci, d->R == 1 && d->S == 1 && d->stride == 1, !d->stem).bm);
}

// Diagnostic: the block tile frx_conv_fwd (dgrad = 0) or frx_conv_dgrad* (dgrad = 1) launches for this layer.
extern "C" int frx_conv_tile(const frx_conv_desc* d, int dgrad, int* bm, int* bn) {
  if (int rc = check_conv(d)) return rc;
  FRX_CHECK_ARG(bm && bn, "conv_tile: NULL pointer");
  const bool pw = d->R == 1 && d->S == 1 && d->stride == 1;
  TileCfg c;
  if (dgrad) {
    const bool s2c = d->stride == 2 && (d->R > 1 || d->S > 1);
    c = pick_tile((long)d->N * d->Hi * d->Wi, d->Ci, (long)d->R * d->S * d->Co, pw, !s2c);
    if (d->Ci % c.bn != 0) c.bn = 64;
  } else {
    c = pick_tile((long)d->N * d->Ho * d->Wo, d->Co, (long)d->R * d->S * d->Ci, pw, !d->stem);
    if (d->Co % c.bn != 0) c.bn = 64;
  }
  *bm = c.bm; *bn = c.bn;
  return FRX_OK;
}

// Diagnostic: the row tile (128 or 64 pixels; 0: not eligible) if the layer's geometry puts frx_conv_fwd* (dgrad = 0) / frx_conv_dgrad_bn* (dgrad = 1) on the patch-mode
// 3x3 kernel when the call has a prologue and no addend (and, with per-tile partial statistics, a 128-pixel row tile).
extern "C" int frx_conv_patch_mode(const frx_conv_desc* d, int dgrad) {
  if (check_conv(d) != FRX_OK) return -1;
  if (d->stem) return 0;
  if (!p3_geometry(d->dtype, d->R, d->S, d->stride, d->pad, dgrad ? d->Wo : d->Wi, dgrad ? d->Co : d->Ci)) return 0;
  int bm, bn;
  p3_tile((long)d->N * d->Ho * d->Wo, dgrad ? d->Ci : d->Co, &bm, &bn);
  return bm;
}

// Diagnostic: the k_igemm instantiation the calling thread's last frx_conv_fwd* / frx_conv_dgrad* launched --
// {BM, BN, waves, K-chunk bytes, LDS-DMA stages (0: register ring), MODE (0 fwd, 1 dgrad, 2 stem, 3 / 4 patch-mode 3x3 fwd /
// dgrad), PRO (0 none, 1 BN+ReLU, 2 BN backward), EPI (0 plain, 1 statistics, 2 / 4 masked BN-backward statistics, 3 fc),
// addend, persistent, staging waves, contraction length}: what bench.py's per-class roofline table is keyed by.
extern "C" int frx_last_conv_launch(int* fields12) {
  FRX_CHECK_ARG(fields12 != nullptr, "last_conv_launch: NULL pointer");
  memcpy(fields12, t_last_launch, sizeof(t_last_launch));
  return FRX_OK;
}

extern "C" int frx_stem_padded_dims(int Hi, int Wi, int* Hp, int* Wp) {
  FRX_CHECK_ARG(Hp && Wp && Hi > 0 && Wi > 0, "stem_padded_dims: bad args");
  // 3 pixels of zero border on each side; the 8-tap (32-element) row read of the last output
  // pixel ends at column 2*(Wo-1)+7, so round the width up to an even count that covers it.
  const int Wo = (Wi + 6 - 7) / 2 + 1, Ho = (Hi + 6 - 7) / 2 + 1;
  *Hp = 2 * (Ho - 1) + 7 > Hi + 6 ? 2 * (Ho - 1) + 7 : Hi + 6;
  int w = 2 * (Wo - 1) + 8;
  if (w < Wi + 6) w = Wi + 6;
  *Wp = (w + 1) & ~1;
  return FRX_OK;
}

static int conv_fwd_impl(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x,
                         const void* w, const float* in_scale, const float* in_shift, int in_relu,
                         const float* bias, void* y, int out_f32, float* stat_partial, const frx_bn_tot* in_bn,
                         float* stat_totals, int stat_replicas, void* x_norm_out = nullptr) {
  if (int rc = check_conv(d)) return rc;
  if (x_norm_out) {
    FRX_CHECK_ARG(in_scale || in_bn, "conv_fwd_keep: x_norm_out is the prologue's output: needs in_scale / in_shift or in_bn");
    FRX_CHECK_ARG(!d->stem && p3_geometry(d->dtype, d->R, d->S, d->stride, d->pad, d->Wi, d->Ci),
                  "conv_fwd_keep: x_norm_out needs a layer on the patch-mode kernel (frx_conv_patch_mode)");
  }
  if (in_bn) {
    FRX_CHECK_ARG(!d->stem, "conv_fwd_tot: the stem takes the raw image (no prologue)");
    FRX_CHECK_ARG(in_bn->totals && in_bn->gamma && in_bn->beta && frx_pow2(in_bn->replicas) && in_bn->count > 0.f,
                  "conv_fwd_tot: in_bn needs totals / gamma / beta, a power-of-two replica count and count > 0");
    FRX_CHECK_ARG(d->Ci <= 2048, "conv_fwd_tot: BN prologue supports up to 2048 input channels (got %d)", d->Ci);
  }
  FRX_CHECK_ARG(!stat_totals || frx_pow2(stat_replicas), "conv_fwd_tot: stat_replicas must be a power of two");
  FRX_CHECK_ARG(x && w && y, "conv_fwd: NULL pointer");
  FRX_CHECK_ARG(!(d->stem && in_scale), "conv_fwd: the stem takes the raw image (no prologue)");
  FRX_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv_fwd: in_scale/in_shift must come together");
  FRX_CHECK_ARG((bias != nullptr) == (out_f32 != 0), "conv_fwd: bias and fp32 output come together (the fc layer)");
  FRX_CHECK_ARG(!(bias && (stat_partial || in_scale)), "conv_fwd: the fc flavour takes no prologue / statistics");
  FRX_CHECK_ARG(!in_scale || d->Ci <= 2048, "conv_fwd: BN prologue supports up to 2048 input channels (got %d)", d->Ci);
  FRX_ENTER(device);
  ConvArgs a{};
  a.X = x; a.W = w; a.Y = y;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
  a.bias = bias; a.stat_partial = stat_partial; a.out_f32 = out_f32;
  a.in_tot = bn_tot_arg(in_bn); a.stat_tot = stat_totals; a.stat_R = stat_replicas;
  a.dy_out = x_norm_out;
  a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo; a.Ncol = d->Co; a.R = d->R; a.S = d->S;
  a.stride = d->stride; a.pad = d->pad;
  a.M = d->N * d->Ho * d->Wo;
  if (d->stem) {
    int hp, wp;
    frx_stem_padded_dims(d->Hi, d->Wi, &hp, &wp);
    a.mode = MODE_STEM; a.Hx = hp; a.Wx = wp; a.Kc = 4;
  } else {
    a.mode = MODE_FWD; a.Hx = d->Hi; a.Wx = d->Wi; a.Kc = d->Ci;
  }
  return launch_igemm((hipStream_t)stream, a, d->dtype);
}

extern "C" int frx_conv_fwd(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x,
                            const void* w, const float* in_scale, const float* in_shift, int in_relu,
                            const float* bias, void* y, int out_f32, float* stat_partial) {
  return conv_fwd_impl(device, stream, d, x, w, in_scale, in_shift, in_relu, bias, y, out_f32, stat_partial, nullptr, nullptr, 0);
}

extern "C" int frx_conv_fwd_keep(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x, const void* w,
                                 const float* in_scale, const float* in_shift, const frx_bn_tot* in_bn, int in_relu, void* y,
                                 float* stat_partial, float* stat_totals, int stat_replicas, void* x_norm_out) {
  FRX_CHECK_ARG(!(in_scale && in_bn), "conv_fwd_keep: prologue constants as arrays OR as totals");
  FRX_CHECK_ARG(!(stat_partial && stat_totals), "conv_fwd_keep: statistics as partial rows OR as totals");
  return conv_fwd_impl(device, stream, d, x, w, in_scale, in_shift, in_relu, nullptr, y, 0, stat_partial, in_bn, stat_totals, stat_replicas,
                       x_norm_out);
}

// The residual merge of the block BEFORE as the prologue of a 1x1 convolution (replaces that block's frx_block_merge_fwd*
// launch: torchvision Bottleneck.forward's `out += identity; out = relu(out)` feeding the next Bottleneck's conv1,
// backbones.py:16-18): x = relu(bn3(y3) + idn') is evaluated while the tiles are staged, stored once (block_out, mask) by the
// first column of tiles, and never read back by this launch.
extern "C" int frx_conv_fwd_merge(int device, frx_stream_t stream, const frx_conv_desc* d, const void* y3, const void* idn,
                                  const void* w, const float* s3, const float* b3, const float* sd, const float* bd,
                                  const frx_bn_tot* bn3, const frx_bn_tot* bnd, void* block_out, uint8_t* mask, void* y,
                                  float* stat_partial, float* stat_totals, int stat_replicas) {
  if (int rc = check_conv(d)) return rc;
  FRX_CHECK_ARG(!d->stem && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0, "conv_fwd_merge: a 1x1 / stride 1 convolution");
  FRX_CHECK_ARG(y3 && idn && w && y && block_out, "conv_fwd_merge: NULL pointer");
  FRX_CHECK_ARG((s3 != nullptr) != (bn3 != nullptr) && (s3 == nullptr) == (b3 == nullptr), "conv_fwd_merge: bn3 as arrays (s3, b3) OR as totals");
  FRX_CHECK_ARG((sd == nullptr) == (bd == nullptr) && !(sd && bn3) && !(bnd && s3), "conv_fwd_merge: the projection's constants in the same form as bn3's");
  FRX_CHECK_ARG(!(stat_partial && stat_totals), "conv_fwd_merge: statistics as partial rows OR as totals");
  FRX_CHECK_ARG(!stat_totals || frx_pow2(stat_replicas), "conv_fwd_merge: stat_replicas must be a power of two");
  FRX_CHECK_ARG(d->Ci <= 2048, "conv_fwd_merge: up to 2048 input channels (got %d)", d->Ci);
  for (const frx_bn_tot* t : {bn3, bnd})
    if (t) FRX_CHECK_ARG(t->totals && t->gamma && t->beta && frx_pow2(t->replicas) && t->count > 0.f,
                         "conv_fwd_merge: a BatchNorm as totals needs totals / gamma / beta, a power-of-two replica count and count > 0");
  FRX_ENTER(device);
  ConvArgs a{};
  a.X = y3; a.X2 = idn; a.W = w; a.Y = y;
  a.in_scale = s3; a.in_shift = b3; a.id_scale = sd; a.id_shift = bd; a.in_relu = 1;
  a.in_tot = bn_tot_arg(bn3); a.id_tot = bn_tot_arg(bnd);
  a.dy_out = block_out; a.mask_out = mask;
  a.stat_partial = stat_partial; a.stat_tot = stat_totals; a.stat_R = stat_replicas;
  a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo; a.Ncol = d->Co; a.R = 1; a.S = 1; a.stride = 1; a.pad = 0;
  a.M = d->N * d->Ho * d->Wo;
  a.mode = MODE_FWD; a.Hx = d->Hi; a.Wx = d->Wi; a.Kc = d->Ci;
  return launch_igemm((hipStream_t)stream, a, d->dtype);
}

extern "C" int frx_conv_fwd_tot(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x, const void* w,
                                const frx_bn_tot* in_bn, int in_relu, void* y, float* stat_totals, int stat_replicas) {
  return conv_fwd_impl(device, stream, d, x, w, nullptr, nullptr, in_relu, nullptr, y, 0, nullptr, in_bn, stat_totals, stat_replicas);
}

static int dgrad_impl(int device, frx_stream_t stream, const frx_conv_desc* d, const void* dy, const void* w_crsk,
                      const void* addend, void* dx, const frx_dgrad_fuse* f) {
  if (int rc = check_conv(d)) return rc;
  FRX_CHECK_ARG(!d->stem, "conv_dgrad: the stem has no input gradient");
  FRX_CHECK_ARG(dy && w_crsk && dx, "conv_dgrad: NULL pointer");
  ConvArgs a{};
  a.X = dy; a.W = w_crsk; a.Y = dx; a.addend = addend;
  a.N = d->N; a.Hx = d->Ho; a.Wx = d->Wo; a.Kc = d->Co;
  a.Ho = d->Hi; a.Wo = d->Wi; a.Ncol = d->Ci; a.R = d->R; a.S = d->S;
  a.stride = d->stride; a.pad = d->pad;
  a.M = d->N * d->Hi * d->Wi;
  a.mode = MODE_DGRAD;
  a.s2c = (d->stride == 2 && (d->R > 1 || d->S > 1)) ? 1 : 0;
  if (f) {
    if (f->pro_y) {
      FRX_CHECK_ARG((f->pro_coef != nullptr) != (f->pro_tot != nullptr), "conv_dgrad_bn: pro_y needs pro_coef or pro_tot (one of them)");
      FRX_CHECK_ARG(d->Co <= 2048, "conv_dgrad_bn: BN prologue supports up to 2048 channels (got %d)", d->Co);
      a.X2 = f->pro_y;
      if (f->pro_coef) { a.in_scale = f->pro_coef; a.in_shift = f->pro_coef + d->Co; a.pro_gam = f->pro_coef + 2 * d->Co; }
      else {
        const frx_bn_tot* t = f->pro_tot;
        FRX_CHECK_ARG(t->totals && t->gamma && t->mean && t->invstd && frx_pow2(t->replicas) && t->count > 0.f,
                      "conv_dgrad_bn: pro_tot needs totals / gamma / mean / invstd, a power-of-two replica count and count > 0");
        a.pro_tot = bn_tot_arg(t);
      }
    }
    if (f->addend_stride == 2) {
      FRX_CHECK_ARG(addend != nullptr, "conv_dgrad_bn: addend_stride without addend");
      a.add_stride = 2;
    } else {
      FRX_CHECK_ARG(f->addend_stride == 0 || f->addend_stride == 1, "conv_dgrad_bn: addend_stride must be 0, 1 or 2");
    }
    if (f->pro_dy_out) {
      FRX_CHECK_ARG(f->pro_y != nullptr, "conv_dgrad_bn: pro_dy_out needs the BN prologue (pro_y)");
      FRX_CHECK_ARG((d->R == 1 && d->S == 1) || p3_geometry(d->dtype, d->R, d->S, d->stride, d->pad, d->Wo, d->Co),
                    "conv_dgrad_bn: pro_dy_out is for 1x1 convs and the patch-mode 3x3 (each dy element is transformed once)");
      a.dy_out = f->pro_dy_out;
    }
    if (f->epi_y) {
      FRX_CHECK_ARG(f->epi_mean && f->epi_invstd && ((f->epi_partial != nullptr) != (f->epi_totals != nullptr)),
                    "conv_dgrad_bn: epilogue needs mean / invstd and ONE of partial / totals");
      FRX_CHECK_ARG(!f->epi_totals || frx_pow2(f->epi_replicas), "conv_dgrad_bn: epi_replicas must be a power of two");
      FRX_CHECK_ARG(f->epi_out || f->epi_out_bits || (f->epi_scale && f->epi_shift), "conv_dgrad_bn: epilogue mask needs epi_out(_bits) or scale/shift");
      a.epi_bnbwd = 1; a.e_y = f->epi_y; a.e_out = f->epi_out; a.e_scale = f->epi_scale; a.e_shift = f->epi_shift;
      a.e_bits = (const unsigned char*)f->epi_out_bits;
      a.e_mean = f->epi_mean; a.e_invstd = f->epi_invstd; a.stat_partial = f->epi_partial;
      a.stat_tot = f->epi_totals; a.stat_R = f->epi_replicas;
    }
  }
  FRX_ENTER(device);
  return launch_igemm((hipStream_t)stream, a, d->dtype);
}

extern "C" int frx_conv_dgrad(int device, frx_stream_t stream, const frx_conv_desc* d, const void* dy,
                              const void* w_crsk, const void* addend, void* dx) {
  return dgrad_impl(device, stream, d, dy, w_crsk, addend, dx, nullptr);
}

extern "C" int frx_conv_dgrad_bn(int device, frx_stream_t stream, const frx_conv_desc* d, const void* dz,
                                 const void* w_crsk, const void* addend, void* dx, const frx_dgrad_fuse* fuse) {
  FRX_CHECK_ARG(fuse != nullptr, "conv_dgrad_bn: fuse is NULL");
  return dgrad_impl(device, stream, d, dz, w_crsk, addend, dx, fuse);
}

extern "C" int frx_conv_dgrad_stat_rows(const frx_conv_desc* d) {
  if (check_conv(d) != FRX_OK) return -1;
  const long M = (long)d->N * d->Hi * d->Wi;
  const bool s2c = d->stride == 2 && (d->R > 1 || d->S > 1);
  const int bm = pick_tile(M, d->Ci, (long)d->R * d->S * d->Co, d->R == 1 && d->S == 1 && d->stride == 1, !s2c).bm;
  if (s2c) {                                             // parity-class tiles (launch_igemm)
    int t = 0;
    for (int cls = 0; cls < 4; ++cls) t += cdiv((long)d->N * ((d->Hi - (cls >> 1) + 1) / 2) * ((d->Wi - (cls & 1) + 1) / 2), bm);
    return t;
  }
  return cdiv(M, bm);
}

// Geometry shared by the per-layer and the grouped launches; the pixel split is the caller's policy.
struct WgradGeom { int bt, wmode, taps, nchunks, tiles; };
static int wgrad_args(const frx_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, int in_relu,
                      const void* dy, const void* pro_y, const float* pro_coef, float* dw, WgradArgs& a, WgradGeom& g) {
  if (int rc = check_conv(d)) return rc;
  FRX_CHECK_ARG(x && dy && dw, "conv_wgrad: NULL pointer");
  FRX_CHECK_ARG(!(d->stem && in_scale), "conv_wgrad: the stem takes the raw image (no prologue)");
  FRX_CHECK_ARG((pro_y == nullptr) == (pro_coef == nullptr), "conv_wgrad: pro_y / pro_coef come together");
  a = WgradArgs{};
  a.X = x; a.dY = dy; a.dW = dw;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
  a.dY2 = pro_y; a.y_coef = pro_coef;
  a.N = d->N; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.R = d->R; a.S = d->S;
  a.stride = d->stride; a.pad = d->pad; a.M = d->N * d->Ho * d->Wo; a.stem = d->stem;
  g.taps = d->R * d->S;
  int ci_extent = d->Ci;
  if (d->stem) {
    int hp, wp;
    frx_stem_padded_dims(d->Hi, d->Wi, &hp, &wp);
    a.Hx = hp; a.Wx = wp; a.S = 1; g.taps = d->R; ci_extent = 32;
  } else {
    a.Hx = d->Hi; a.Wx = d->Wi;
  }
  const int kp = d->dtype == FRX_BF16 ? 32 : 16;
  g.bt = (d->Co <= 64 || ci_extent <= 64) ? 64 : 128;
  a.tilesCo = cdiv(d->Co, g.bt);
  a.tilesCi = cdiv(ci_extent, g.bt);
  g.nchunks = cdiv(a.M, kp);
  g.tiles = a.tilesCo * a.tilesCi * g.taps;
  const size_t esz = d->dtype == FRX_BF16 ? 2 : 4;
  const size_t xb = (size_t)d->N * a.Hx * a.Wx * (d->stem ? 4 : d->Ci) * esz, yb = (size_t)a.M * d->Co * esz;
  FRX_CHECK_ARG(xb < 0x80000000ull && yb < 0x80000000ull, "wgrad: tensors must stay below 2 GiB (32-bit buffer offsets)");
  a.xbytes = (unsigned)xb; a.ybytes = (unsigned)yb;
  g.wmode = d->stem ? WG_STEM : ((d->R == 1 && d->S == 1 && d->stride == 1) ? WG_POINTWISE : WG_GENERAL);
  a.variant = (g.bt == 128 ? WGV_BT128 : 0) | (g.wmode == WG_GENERAL ? WGV_GENERAL : 0) | (g.wmode == WG_STEM ? WGV_STEM : 0) |
              (in_scale ? WGV_PRO : 0) | (pro_y ? WGV_YPRO : 0);
  return FRX_OK;
}

static int wgrad_impl(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x,
                      const float* in_scale, const float* in_shift, int in_relu, const void* dy,
                      const void* pro_y, const float* pro_coef, float* dw) {
  WgradArgs a; WgradGeom g;
  if (int rc = wgrad_args(d, x, in_scale, in_shift, in_relu, dy, pro_y, pro_coef, dw, a, g)) return rc;
  FRX_ENTER(device);
  // Split the pixel axis so ~2 blocks per CU exist, but keep >= 32 K-chunks per block (below that
  // the 64-atomics-per-lane epilogue dominates); round to a multiple of 8 for the XCD mapping.
  const char* env_b = getenv("FRX_WGRAD_BLOCKS");
  const char* env_c = getenv("FRX_WGRAD_MINCHUNKS");
  const int target_blocks = env_b ? atoi(env_b) : 512, min_chunks = env_c ? atoi(env_c) : 32;   // measured sweep (atomics vs occupancy)
  int splits = cdiv(target_blocks, g.tiles);
  if (splits > g.nchunks / min_chunks) splits = g.nchunks / min_chunks;
  if (splits >= 8) splits = splits / 8 * 8;
  if (splits < 1) splits = 1;
  a.chunks_per_split = cdiv(g.nchunks, splits);
  splits = cdiv(g.nchunks, a.chunks_per_split);
  a.splits = splits;
  return launch_wgrad((hipStream_t)stream, a, d->dtype, g.bt, g.wmode, in_scale != nullptr, pro_y != nullptr, g.tiles * splits);
}

// ---- grouped launch: table = [njobs] WgradArgs, 8 draw counters (one per XCD, zero between launches), then [nitems] WgradItem
static inline size_t group_counters_offset(int njobs) { return round_up((long)njobs * (long)sizeof(WgradArgs), 256); }
static inline size_t group_items_offset(int njobs) { return group_counters_offset(njobs) + 256; }
static int group_chunks_per_item() { return 64; }      // measured sweep (scripts/group_sweep.sh, round 2)

// A job with `gram` / `xsum` set is a DECOMPOSED weight gradient (frx_wgrad_gram_finish): its main layer accumulates the plain
// dz^T x into dw, and a second, internal layer -- the same x as BOTH operands, one Ci x Ci tile -- accumulates G = x^T x into
// `gram` and the column sums of x into `xsum`.  The table therefore holds more layers than there are jobs: *nlayers.
static int gram_args(const frx_wgrad_job& j, WgradArgs& ag, WgradGeom& gg) {
  FRX_CHECK_ARG(j.gram && j.xsum, "wgrad_group: gram and xsum come together");
  FRX_CHECK_ARG(!j.pro_y && !j.pro_coef, "wgrad_group: a decomposed job takes the plain dz (no pro_y / pro_coef)");
  FRX_CHECK_ARG(j.d.R == 1 && j.d.S == 1 && j.d.stride == 1 && !j.d.stem, "wgrad_group: decomposed jobs are 1x1 / stride 1 convolutions");
  FRX_CHECK_ARG(j.d.Ci <= 128, "wgrad_group: a decomposed job's x^T x must fit one tile (Ci <= 128, got %d)", j.d.Ci);
  frx_conv_desc dg = j.d;
  dg.Co = dg.Ci;
  if (int rc = wgrad_args(&dg, j.x, j.in_scale, j.in_shift, j.in_relu, j.x, nullptr, nullptr, j.gram, ag, gg)) return rc;
  ag.variant |= WGV_GRAM;
  ag.xsum = j.xsum;
  return FRX_OK;
}

static int group_build(const frx_wgrad_job* jobs, int njobs, std::vector<WgradArgs>* layers, std::vector<WgradItem>* items, int* nlayers) {
  FRX_CHECK_ARG(jobs && njobs > 0, "wgrad_group: no jobs");
  int next_layer = 0;
  const int target = group_chunks_per_item();
  const int scatter_chunks = 128;
  // (layer, split) groups stream the same pixel range: keep each on ONE XCD (block b runs on XCD b % 8, item p is
  // taken by block p % grid), so its tiles share that L2.  Groups go round-robin to the least loaded XCD list.
  std::vector<std::vector<WgradItem>> xl(8);
  long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int li = 0; li < njobs; ++li) {
    const frx_wgrad_job& j = jobs[li];
    FRX_CHECK_ARG(j.d.dtype == jobs[0].d.dtype, "wgrad_group: all jobs must share a dtype");
    WgradArgs a; WgradGeom g;
    if (int rc = wgrad_args(&j.d, j.x, j.in_scale, j.in_shift, j.in_relu, j.dy, j.pro_y, j.pro_coef, j.dw, a, g)) return rc;
    int splits = (g.nchunks + target / 2) / target;
    if (splits < 1) splits = 1;
    a.chunks_per_split = cdiv(g.nchunks, splits);
    splits = cdiv(g.nchunks, a.chunks_per_split);
    a.splits = splits;
    if (layers) layers->push_back(a);
    const int lmain = next_layer++;
    int lgram = -1;
    if (j.gram || j.xsum) {
      WgradArgs ag; WgradGeom gg;
      if (int rc = gram_args(j, ag, gg)) return rc;
      ag.chunks_per_split = a.chunks_per_split; ag.splits = splits;      // the same pixel ranges as the main layer's splits
      if (layers) layers->push_back(ag);
      lgram = next_layer++;
    }
    const long cost = (long)a.chunks_per_split * (g.bt == 128 ? 4 : 1);
    for (int sp = 0; sp < splits; ++sp) {
      int best = 0;
      for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
      if (lgram >= 0) { xl[best].push_back(WgradItem{lgram, sp, 0, 0}); load[best] += cost; }      // (next to the tiles that stream the same x)
      // ... unless the layer has few pixels and many tiles (layer4: 4096 pixels, up to 144 tiles per split): its operands
      // are small and L2-resident anyway, and a whole (layer, split) group on one list leaves the eight lists unbalanced
      // (the launch ends with the longest).  Those tiles go round-robin over the XCDs (upper list 563 -> 513 us).
      if (g.nchunks <= scatter_chunks) {
        int q = 0;
        for (int tap = 0; tap < g.taps; ++tap)
          for (int t = 0; t < a.tilesCo * a.tilesCi; ++t, ++q) { xl[(best + q) & 7].push_back(WgradItem{lmain, sp, t, tap}); load[(best + q) & 7] += cost; }
        continue;
      }
      for (int tap = 0; tap < g.taps; ++tap)
        for (int t = 0; t < a.tilesCo * a.tilesCi; ++t) xl[best].push_back(WgradItem{lmain, sp, t, tap});
      load[best] += cost * g.tiles;
    }
  }
  if (nlayers) *nlayers = next_layer;
  size_t longest = 0;
  for (auto& v : xl) longest = v.size() > longest ? v.size() : longest;
  if (items) {
    for (size_t k = 0; k < longest; ++k)
      for (int x = 0; x < 8; ++x)
        items->push_back(k < xl[x].size() ? xl[x][k] : WgradItem{-1, 0, 0, 0});     // -1: padding, skipped by the kernel
  }
  return (int)(longest * 8);
}

extern "C" int64_t frx_wgrad_group_bytes(const frx_wgrad_job* jobs, int njobs) {
  int nl = 0;
  const int n = group_build(jobs, njobs, nullptr, nullptr, &nl);
  if (n < 0) return n;
  return (int64_t)group_items_offset(nl) + (int64_t)n * (int64_t)sizeof(WgradItem);
}

extern "C" int frx_wgrad_group_plan(int device, frx_stream_t stream, const frx_wgrad_job* jobs, int njobs, void* table_host,
                                    void* table_dev, int64_t table_bytes, int* nitems, int* small_tiles, int* nlayers) {
  FRX_CHECK_ARG(table_host && table_dev && nitems && small_tiles && nlayers, "wgrad_group_plan: NULL pointer");
  std::vector<WgradArgs> layers; std::vector<WgradItem> items;
  const int n = group_build(jobs, njobs, &layers, &items, nlayers);
  if (n < 0) return n;
  const size_t off = group_items_offset(*nlayers), need = off + items.size() * sizeof(WgradItem);
  FRX_CHECK_ARG((size_t)table_bytes >= need, "wgrad_group_plan: table needs %zu bytes, got %ld", need, (long)table_bytes);
  FRX_ENTER(device);
  // host image first, then ONE stream-ordered copy: nothing here synchronises, and the copy is ordered against whatever
  // the caller's stream did to table_dev before (round 1 used a blocking hipMemcpy, which raced torch's non-blocking
  // streams: an earlier zero-fill of the table could land after it)
  memset(table_host, 0, need);
  memcpy(table_host, layers.data(), layers.size() * sizeof(WgradArgs));
  memcpy((char*)table_host + off, items.data(), items.size() * sizeof(WgradItem));
  FRX_HIP(hipMemcpyAsync(table_dev, table_host, need, hipMemcpyHostToDevice, (hipStream_t)stream));
  *nitems = (int)items.size();
  *small_tiles = 1;
  for (const WgradArgs& a : layers) if (a.variant & WGV_BT128) *small_tiles = 0;
  return FRX_OK;
}

extern "C" int frx_wgrad_group_run(int device, frx_stream_t stream, int dtype, void* table_dev, int nlayers, int nitems,
                                   int small_tiles) {
  FRX_CHECK_ARG(dtype == FRX_F32 || dtype == FRX_BF16, "wgrad_group_run: dtype");
  FRX_CHECK_ARG(table_dev && nlayers > 0 && nitems > 0, "wgrad_group_run: bad args");
  FRX_ENTER(device);
  return launch_wgrad_grouped((hipStream_t)stream, dtype, (const WgradArgs*)table_dev,
                              (const WgradItem*)((const char*)table_dev + group_items_offset(nlayers)), nitems, small_tiles != 0,
                              (int*)((char*)table_dev + group_counters_offset(nlayers)));
}

// ---- closing a DECOMPOSED weight gradient: dW = alpha (.) A + beta (.) (W G) + gam (x) s
// For a 1x1 conv whose output gradient is the BatchNorm backward dy = alpha*dz + beta*y + gam of (dz, y), y = x W^T:
//   dW[co][ci] = sum_m dy[m,co] x[m,ci] = alpha_co * A[co][ci] + beta_co * sum_cj W[co][cj] G[cj][ci] + gam_co * s[ci]
// with A = dz^T x (in dw already), G = x^T x and s = column sums of x (frx_wgrad_group: `gram`, `xsum`).  One launch for
// every listed layer; the last block of a layer to finish zeroes that layer's G and s for the next step.
namespace frx {
template <typename T>
__global__ __launch_bounds__(256) void k_wgrad_gram_finish(const int64_t* __restrict__ table, int n) {
  int e = 0;
  for (int i = 1; i < n; ++i) e = ((int64_t)blockIdx.x >= table[i * 8 + 7]) ? i : e;
  const int64_t* t = table + e * 8;
  float* dw = (float*)t[0];
  const T* w = (const T*)t[1];
  float* G = (float*)t[2];
  float* s = (float*)t[3];
  const float* coef = (const float*)t[4];
  const int Co = (int)t[5], Ci = (int)(t[6] & 0xffffffff);
  int* counter = (int*)(G + (long)Ci * Ci);                 // (one int behind the layer's G: frx_wgrad_gram_bytes)
  const int local = (int)((int64_t)blockIdx.x - t[7]);
  const long idx = (long)local * 256 + threadIdx.x;
  if (idx < (long)Co * Ci) {
    const int co = (int)(idx / Ci), ci = (int)(idx - (long)co * Ci);
    float acc = 0.f;
    for (int cj = 0; cj < Ci; ++cj) acc = fmaf((float)w[(long)co * Ci + cj], G[(long)cj * Ci + ci], acc);
    dw[idx] = fmaf(coef[co], dw[idx], fmaf(coef[Co + co], acc, coef[2 * Co + co] * s[ci]));
  }
  // the layer's last block zeroes G and s (every block has consumed them: the loads above feed the store)
  __shared__ int last;
  __syncthreads();
  const int nblk = (Co * Ci + 255) / 256;
  if (threadIdx.x == 0) {
    const int k = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (k == nblk - 1);
    if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (last) {
    for (int i = threadIdx.x; i < Ci * Ci; i += 256) G[i] = 0.f;
    for (int i = threadIdx.x; i < Ci; i += 256) s[i] = 0.f;
  }
}
}  // namespace frx

extern "C" int frx_wgrad_gram_finish(int device, frx_stream_t stream, int dtype, int n, const int64_t* table_dev, int total_blocks) {
  FRX_CHECK_ARG(dtype == FRX_F32 || dtype == FRX_BF16, "wgrad_gram_finish: dtype");
  FRX_CHECK_ARG(n > 0 && table_dev && total_blocks > 0, "wgrad_gram_finish: bad args");
  FRX_ENTER(device);
  if (dtype == FRX_BF16) hipLaunchKernelGGL(k_wgrad_gram_finish<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
  else hipLaunchKernelGGL(k_wgrad_gram_finish<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

extern "C" int frx_conv_wgrad(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x,
                              const float* in_scale, const float* in_shift, int in_relu, const void* dy,
                              float* dw) {
  return wgrad_impl(device, stream, d, x, in_scale, in_shift, in_relu, dy, nullptr, nullptr, dw);
}

extern "C" int frx_conv_wgrad_bn(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x,
                                 const float* in_scale, const float* in_shift, int in_relu, const void* dz,
                                 const void* pro_y, const float* pro_coef, float* dw) {
  FRX_CHECK_ARG(pro_y && pro_coef, "conv_wgrad_bn: pro_y / pro_coef are NULL");
  FRX_CHECK_ARG(!d || !d->stem, "conv_wgrad_bn: not for the stem");
  return wgrad_impl(device, stream, d, x, in_scale, in_shift, in_relu, dz, pro_y, pro_coef, dw);
}
