// k_igemm instantiations: the patch-mode 3x3 convolutions (forward with the BN+ReLU prologue, input gradient with the
// BN-backward prologue and the masked-statistics epilogue).
#include "conv_launch.h"
namespace frx {
#define FRX_P3(BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, NS_) \
  do { note_igemm_launch(BM_, BN_, WM_ * WN_, 64, NS_, MODE_, PRO_, EPI_, 0, 0, 0); \
  hipLaunchKernelGGL((k_igemm<bf16_t, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, false, 64, 3, NS_, false>), dim3(a.nvb), dim3(256), igemm_pro_lds(PRO_, a.Kc), st, a); } while (0)
// tiles: 128 x 128 and 128 x 64 on 2 x 2 waves; 64 x 128 on 1 x 4 waves for launches that would otherwise leave CUs without a
// tile (layer4: 4096 pixels).  Weight stages: what two blocks per CU -- three on the 64-column tile -- leave of the 160 KB
// next to the patch buffers.
#define FRX_P3_T(MODE_, PRO_, EPI_, NS128_, NS64_)                                         \
  do {                                                                                     \
    if (bm == 64) FRX_P3(64, 128, 1, 4, MODE_, PRO_, EPI_, NS128_);                         \
    else if (bn == 128) FRX_P3(128, 128, 2, 2, MODE_, PRO_, EPI_, NS128_);                  \
    else FRX_P3(128, 64, 2, 2, MODE_, PRO_, EPI_, NS64_);                                   \
  } while (0)
// the same tiles with staging waves (SPEC: 512 threads, one block per CU, 7 weight stages) for launches of at most 256 tiles
#define FRX_P3S(BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_) \
  do { note_igemm_launch(BM_, BN_, WM_ * WN_, 64, 7, MODE_, PRO_, EPI_, 0, 0, 1); \
  hipLaunchKernelGGL((k_igemm<bf16_t, BM_, BN_, WM_, WN_, MODE_, PRO_, EPI_, false, 64, 3, 7, false, true>), dim3(a.nvb), dim3(512), igemm_pro_lds(PRO_, a.Kc), st, a); } while (0)
#define FRX_P3S_T(MODE_, PRO_, EPI_)                                                       \
  do {                                                                                     \
    if (bm == 64) FRX_P3S(64, 128, 1, 4, MODE_, PRO_, EPI_);                                \
    else FRX_P3S(128, 128, 2, 2, MODE_, PRO_, EPI_);                                        \
  } while (0)
int launch_igemm_p3(hipStream_t st, const ConvArgs& a, int epi, int bm, int bn) {
  const bool spec = bn == 128 && (long)a.tilesM * a.tilesN <= 256;
  if (spec && a.mode == MODE_FWD && epi == EPI_STATS) FRX_P3S_T(MODE_FWD3, 1, EPI_STATS);
  else if (spec && a.mode == MODE_FWD && epi == EPI_PLAIN) FRX_P3S_T(MODE_FWD3, 1, EPI_PLAIN);
  else if (spec && a.mode == MODE_DGRAD && epi == EPI_BNBWD && a.X2) FRX_P3S_T(MODE_DGRAD3, 2, EPI_BNBWD);
  else if (a.mode == MODE_FWD) {
    if (epi == EPI_STATS) FRX_P3_T(MODE_FWD3, 1, EPI_STATS, 5, 5);
    else if (epi == EPI_PLAIN) FRX_P3_T(MODE_FWD3, 1, EPI_PLAIN, 5, 5);
    else { set_error("igemm p3 fwd: unsupported epilogue %d", epi); return FRX_ERR_ARG; }
  } else {
    if (epi == EPI_BNBWD && a.X2) FRX_P3_T(MODE_DGRAD3, 2, EPI_BNBWD, 4, 4);
    else if (epi == EPI_BNBWD && bn == 64) FRX_P3(128, 64, 2, 2, MODE_DGRAD3, 0, EPI_BNBWD, 6);      // materialised dy (layer1: see conv.hip)
    else { set_error("igemm p3 dgrad: unsupported epilogue %d", epi); return FRX_ERR_ARG; }
  }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
}  // namespace frx

FRX_DBG_EXPORT(frx_debug_times_p3)
