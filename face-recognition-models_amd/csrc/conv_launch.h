// Launch plumbing shared by the convolution translation units (the kernel templates are instantiated
// in several .hip files so that `make -j` compiles them in parallel).
#pragma once
#include "conv_kernels.h"
#include <stdlib.h>

namespace frx {

struct TileCfg { int bm, bn; };

static inline TileCfg pick_tile(long M, int Ncol) {
  if (const char* e = getenv("FRX_IGEMM_TILE")) {           // tuning aid: "128x128" | "128x64" | "64x64"
    if (e[0] == '6') return {64, 64};
    if (e[0] == '1' && e[4] == '6') return {128, 64};
    if (e[0] == '1' && Ncol % 128 == 0) return {128, 128};
  }
  if (Ncol <= 64) return {128, 64};
  // measured per ResNet-50 shape (scripts/tile_sweep.py): the square 128x128 tile wins down to ~3/4 of a wave of
  // blocks (twice the MFMA work per staged byte and per prologue evaluation); below that the 64x64 tile's 4x
  // block count beats the 128x64 one's 2x.  The 128x128 tile runs EIGHT waves (2x4, 64x32 each): half the
  // staging work and accumulators per wave keep it under 128 registers, i.e. 16 waves per CU instead of 8 --
  // 5-15 % on the forward convs, 3-10 % on the dgrads over four 64x64 waves.
  const long mt128 = (M + 127) / 128;
  if (mt128 * ((Ncol + 127) / 128) >= 192) return {128, 128};
  return {64, 64};
}

// each returns FRX_OK / FRX_ERR_*; `a` is completed (tiles, byte sizes) by the caller
int launch_igemm_fwd(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int pro, int epi);
int launch_igemm_stem(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi);
int launch_igemm_dgrad_plain(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi, bool add);
int launch_igemm_dgrad_bn(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi, bool add);

int launch_wgrad(hipStream_t st, const WgradArgs& a, int dtype, int bt, int wmode, bool pro, bool ypro, int grid);
int launch_wgrad_grouped(hipStream_t st, int dtype, const WgradArgs* layers, const WgradItem* items, int nitems);

#define FRX_IGEMM_LAUNCH(T_, MODE_, PRO_, EPI_, ADD_)                                                                     \
  do {                                                                                                                    \
    if (c.bm == 128 && c.bn == 128) hipLaunchKernelGGL((k_igemm<T_, 128, 128, 2, 4, MODE_, PRO_, EPI_, ADD_>), dim3(grid), dim3(512), 0, st, a); \
    else if (c.bm == 128 && c.bn == 64) hipLaunchKernelGGL((k_igemm<T_, 128, 64, 2, 2, MODE_, PRO_, EPI_, ADD_>), dim3(grid), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((k_igemm<T_, 64, 64, 2, 2, MODE_, PRO_, EPI_, ADD_>), dim3(grid), dim3(256), 0, st, a);       \
  } while (0)
#define FRX_IGEMM_DT(MODE_, PRO_, EPI_, ADD_)                                   \
  do {                                                                          \
    if (dtype == FRX_BF16) FRX_IGEMM_LAUNCH(bf16_t, MODE_, PRO_, EPI_, ADD_);   \
    else FRX_IGEMM_LAUNCH(float, MODE_, PRO_, EPI_, ADD_);                      \
  } while (0)

}  // namespace frx
