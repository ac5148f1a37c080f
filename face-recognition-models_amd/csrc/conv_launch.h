// Launch plumbing shared by the convolution translation units (the kernel templates are instantiated
// in several .hip files so that `make -j` compiles them in parallel).
#pragma once
#include "conv_kernels.h"
#include <stdlib.h>
#include <stdio.h>

namespace frx {

struct TileCfg { int bm, bn, waves, kc; int ns = 0; };       // block tile (pixels x channels), waves per block, K-chunk bytes; ns > 0: LDS-DMA staging, ns LDS stages

// LDS-DMA staging (k_igemm<..., NS>) exists for the prologue-free launches on the two production tiles; FRX_IGEMM_DMA=0
// switches it off, =3 / =4 picks the stage count of the 128x128 tile (tuning aid, read per launch)
// Stage count: a chunk issued at step k is consumed at step k + NS - 2, so NS - 2 steps of MFMA work cover its flight.
// In a training step the operands come from HBM / the Infinity Cache (~1-2 us under load, 10x a step's MFMA time): four
// stages (what two blocks per CU can hold) hide a third of it, so launches with at most one block per CU take the deep
// ring (8 x 16 KB / 6 x 24 KB of the CU's 160 KB).  FRX_IGEMM_DMA=0: off; =N: force N stages where instantiated.
static inline int dma_stages(const TileCfg& c, bool prologue_free, bool stem, long tiles) {
  if (!prologue_free || stem) return 0;
  const bool t128 = c.bm == 128 && c.bn == 128 && c.kc == 64, t64 = c.bm == 64 && c.bn == 128 && c.kc == 128;
  if (!t128 && !t64) return 0;
  int ns = tiles <= 256 ? (t128 ? 8 : 6) : (t128 ? 4 : 0);      // (64x128 with more than a block per CU: three stages hide too little -- register ring)
  if (const char* e = getenv("FRX_IGEMM_DMA")) {
    const int v = atoi(e);
    if (v == 0) return 0;
    if (t128 && (v == 4 || v == 8)) ns = v;
    if (t64 && (v == 3 || v == 6)) ns = v;
  }
  return ns;
}

// tuning aid: FRX_IGEMM_TILE = "BMxBNxWAVESxKC" out of the list FRX_IGEMM_LAUNCH instantiates
static inline bool tile_from_env(TileCfg* c) {
  const char* e = getenv("FRX_IGEMM_TILE");
  if (!e) return false;
  int bm, bn, w, kc;
  if (sscanf(e, "%dx%dx%dx%d", &bm, &bn, &w, &kc) != 4) return false;
  *c = TileCfg{bm, bn, w, kc, 0};
  return true;
}

// Block tile of one implicit-GEMM launch: M x Ncol outputs, contraction K = taps x channels.  `pointwise`: 1x1, stride 1
// (rows are whole pixels, no gather); `kc128_ok`: the launch can take 128-byte K-chunks (not the stem, not the
// parity-class input gradients).  A pure function of the layer geometry: frx_conv_stat_rows / frx_conv_dgrad_stat_rows
// (the partial-statistics rows = M tiles) go through it as well.  Measured per ResNet-50 shape on whole training steps
// (scripts/layer_times.py under FRX_IGEMM_TILE, scripts/lt_compare.py).
static inline TileCfg pick_tile(long M, int Ncol, long K, bool pointwise, bool kc128_ok) {
  TileCfg c;
  if (tile_from_env(&c) && Ncol % c.bn == 0 && !(!kc128_ok && c.kc != 64)) return c;
  if (Ncol <= 64) return {128, 64, 4, 64};
  const long t64x128 = (kc128_ok && Ncol % 128 == 0) ? ((M + 63) / 64) * (Ncol / 128) : 0;
  // deep pointwise contractions over few column tiles (layer3/4 conv1 forward, conv3 input gradients, layer4's
  // projection): 64 pixels x 128 channels on four waves side by side with 128-byte K-chunks -- whole 128-byte lines of
  // the activation rows, every wave sharing the one pixel tile (e.g. 256<-1024 at 14x14: 36 -> 31 us, 512<-2048 at
  // 7x7: 48 -> 35 us, 1024->256 forward: 22.9 -> 20.4 us)
  if (pointwise && K >= 1024 && t64x128 >= 192 && t64x128 <= 512) return {64, 128, 4, 128};
  // otherwise the square 128x128 tile wins down to ~3/4 of a wave of blocks (twice the MFMA work per staged byte and
  // per prologue evaluation).  It runs EIGHT waves (2x4, 64x32 each): half the staging work and accumulators per wave
  // keep it under 128 registers, i.e. 16 waves per CU instead of 8.
  const long mt128 = (M + 127) / 128;
  if (mt128 * ((Ncol + 127) / 128) >= 192) return {128, 128, 8, 64};
  // fewer tiles than that (layer4, M = 4096): the 64x128 tile again (3x3 forward 57 -> 48 us, 3x3 input gradient 50 -> 46 us)
  if (t64x128 >= 192) return {64, 128, 4, 128};
  return {64, 64, 4, 64};
}

// each returns FRX_OK / FRX_ERR_*; `a` is completed (tiles, byte sizes) by the caller
int launch_igemm_fwd(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int pro, int epi);
int launch_igemm_stem(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi);
int launch_igemm_dgrad_plain(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi, bool add);
int launch_igemm_dgrad_bn(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi, bool add);

// row-resident pointwise kernels (pw_rows.hip): conv1-type input gradients with the masked-statistics epilogue, bf16
bool pw_rows_dgrad_ok(const ConvArgs& a, int dtype, int epi);
int launch_pw_rows_dgrad(hipStream_t st, const ConvArgs& a);
bool pw_rows_fwd_ok(const ConvArgs& a, int dtype, int epi);       // conv3-type forwards (BN + ReLU prologue, statistics into totals), bf16
int launch_pw_rows_fwd(hipStream_t st, const ConvArgs& a);

// streamed pointwise kernels (pw_stream.hip): conv1-type forwards / conv3-type input gradients of layer1 and layer2, bf16
bool pw_stream_ok(const ConvArgs& a, int dtype, int epi);
int launch_pw_stream(hipStream_t st, const ConvArgs& a);

// patch-mode 3x3 (k_igemm MODE_FWD3 / MODE_DGRAD3: bf16, stride 1, pad 1; 128 x bn tile on 2 x 2 waves, or 64 x 128 on 1 x 4)
int launch_igemm_p3(hipStream_t st, const ConvArgs& a, int epi, int bm, int bn);

int launch_wgrad(hipStream_t st, const WgradArgs& a, int dtype, int bt, int wmode, bool pro, bool ypro, int grid);
int launch_wgrad_grouped(hipStream_t st, int dtype, const WgradArgs* layers, const WgradItem* items, int nitems, bool small_tiles,
                         int* draw_counters);

// diagnostic (frx_last_conv_launch): the template arguments of the k_igemm instantiation this thread launched last
void note_igemm_launch(int bm, int bn, int waves, int kc, int ns, int mode, int pro, int epi, int add, int persist, int spec);

#define FRX_IGEMM_KNP(T_, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, ADD_, KC_, NS_, PERSIST_) \
  do { note_igemm_launch(BM_, BN_, WM_ * WN_, KC_, NS_, MODE_, PRO_, EPI_, ADD_, PERSIST_, 0); \
  hipLaunchKernelGGL((k_igemm<T_, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, ADD_, KC_, 3, NS_, PERSIST_>), dim3(PERSIST_ ? grid : a.nvb), dim3(64 * WM_ * WN_), igemm_pro_lds(PRO_, a.Kc), st, a); } while (0)
#define FRX_IGEMM_KN(T_, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, ADD_, KC_, NS_) FRX_IGEMM_KNP(T_, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, ADD_, KC_, NS_, false)
#define FRX_IGEMM_K(T_, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, ADD_, KC_) FRX_IGEMM_KN(T_, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, ADD_, KC_, 0)
#define FRX_IGEMM_LAUNCH64(T_, MODE_, PRO_, EPI_, ADD_)                                                                    \
  do {                                                                                                                     \
    if (c.bm == 128 && c.bn == 128 && grid < a.nvb && MODE_ == MODE_FWD) FRX_IGEMM_KNP(T_, 128, 128, 2, 4, MODE_, PRO_, EPI_, ADD_, 64, 0, (MODE_ == MODE_FWD)); \
    else if (c.bm == 128 && c.bn == 64 && grid < a.nvb && MODE_ == MODE_FWD) FRX_IGEMM_KNP(T_, 128, 64, 2, 2, MODE_, PRO_, EPI_, ADD_, 64, 0, (MODE_ == MODE_FWD)); \
    else if (c.bm == 128 && c.bn == 128) FRX_IGEMM_K(T_, 128, 128, 2, 4, MODE_, PRO_, EPI_, ADD_, 64);                      \
    else if (c.bm == 128 && c.bn == 64) FRX_IGEMM_K(T_, 128, 64, 2, 2, MODE_, PRO_, EPI_, ADD_, 64);                        \
    else FRX_IGEMM_K(T_, 64, 64, 2, 2, MODE_, PRO_, EPI_, ADD_, 64);                                                        \
  } while (0)
#define FRX_IGEMM_LAUNCH(T_, MODE_, PRO_, EPI_, ADD_)                                                                      \
  do {                                                                                                                     \
    if (c.kc == 128 && c.bm == 64 && c.bn == 128) FRX_IGEMM_K(T_, 64, 128, 1, 4, MODE_, PRO_, EPI_, ADD_, 128);             \
    else FRX_IGEMM_LAUNCH64(T_, MODE_, PRO_, EPI_, ADD_);                                                                  \
  } while (0)
#define FRX_IGEMM_DT(MODE_, PRO_, EPI_, ADD_)                                   \
  do {                                                                          \
    if (dtype == FRX_BF16) FRX_IGEMM_LAUNCH(bf16_t, MODE_, PRO_, EPI_, ADD_);   \
    else FRX_IGEMM_LAUNCH(float, MODE_, PRO_, EPI_, ADD_);                      \
  } while (0)

// prologue-free launches: the LDS-DMA instantiations where c.ns says so
#define FRX_IGEMM_LAUNCH_DMA(T_, MODE_, EPI_, ADD_)                                                                        \
  do {                                                                                                                     \
    if (c.ns == 4 && c.bm == 128 && c.bn == 128) FRX_IGEMM_KN(T_, 128, 128, 2, 4, MODE_, 0, EPI_, ADD_, 64, 4);            \
    else if (c.ns == 8 && c.bm == 128 && c.bn == 128) FRX_IGEMM_KN(T_, 128, 128, 2, 4, MODE_, 0, EPI_, ADD_, 64, 8);       \
    else if (c.ns == 3 && c.bm == 64 && c.bn == 128 && c.kc == 128) FRX_IGEMM_KN(T_, 64, 128, 1, 4, MODE_, 0, EPI_, ADD_, 128, 3); \
    else if (c.ns == 6 && c.bm == 64 && c.bn == 128 && c.kc == 128) FRX_IGEMM_KN(T_, 64, 128, 1, 4, MODE_, 0, EPI_, ADD_, 128, 6); \
    else FRX_IGEMM_LAUNCH(T_, MODE_, 0, EPI_, ADD_);                                                                       \
  } while (0)
#define FRX_IGEMM_DT_DMA(MODE_, EPI_, ADD_)                                     \
  do {                                                                          \
    if (dtype == FRX_BF16) FRX_IGEMM_LAUNCH_DMA(bf16_t, MODE_, EPI_, ADD_);     \
    else FRX_IGEMM_LAUNCH_DMA(float, MODE_, EPI_, ADD_);                        \
  } while (0)

// (the stem's rows are 64 bytes per tap row: 64-byte K-chunks only)
#define FRX_IGEMM_DT64(MODE_, PRO_, EPI_, ADD_)                                   \
  do {                                                                            \
    if (dtype == FRX_BF16) FRX_IGEMM_LAUNCH64(bf16_t, MODE_, PRO_, EPI_, ADD_);   \
    else FRX_IGEMM_LAUNCH64(float, MODE_, PRO_, EPI_, ADD_);                      \
  } while (0)

}  // namespace frx
