// k_igemm instantiations: dgrad with PRO = 0.
#include "conv_launch.h"
namespace frx {
int launch_igemm_dgrad_plain(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi, bool add) {
  if (add) {
    if (epi == EPI_PLAIN) FRX_IGEMM_DT_DMA(MODE_DGRAD, EPI_PLAIN, true);
    else if (epi == EPI_BNBWD) FRX_IGEMM_DT_DMA(MODE_DGRAD, EPI_BNBWD, true);
    else FRX_IGEMM_DT_DMA(MODE_DGRAD, EPI_BNBWD_OUT, true);
  } else {
    if (epi == EPI_PLAIN) FRX_IGEMM_DT_DMA(MODE_DGRAD, EPI_PLAIN, false);
    else if (epi == EPI_BNBWD) FRX_IGEMM_DT_DMA(MODE_DGRAD, EPI_BNBWD, false);
    else FRX_IGEMM_DT_DMA(MODE_DGRAD, EPI_BNBWD_OUT, false);
  }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
}  // namespace frx

FRX_DBG_EXPORT(frx_debug_times_dgrad_plain)
