// k_igemm instantiations: forward (and fc) and the stem.
#include "conv_launch.h"
namespace frx {
int launch_igemm_fwd_merge(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi);      // igemm_fwd_merge.hip
int launch_igemm_fwd(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int pro, int epi) {
  if (pro == 3) return launch_igemm_fwd_merge(st, a, dtype, c, grid, epi);
  if (pro == 1) {
    if (epi == EPI_STATS) FRX_IGEMM_DT(MODE_FWD, 1, EPI_STATS, false);
    else if (epi == EPI_PLAIN) FRX_IGEMM_DT(MODE_FWD, 1, EPI_PLAIN, false);
    else { set_error("igemm fwd: unsupported epilogue %d with a prologue", epi); return FRX_ERR_ARG; }
  } else {
    if (epi == EPI_STATS) FRX_IGEMM_DT_DMA(MODE_FWD, EPI_STATS, false);
    else if (epi == EPI_PLAIN) FRX_IGEMM_DT_DMA(MODE_FWD, EPI_PLAIN, false);
    else if (epi == EPI_FC) FRX_IGEMM_DT(MODE_FWD, 0, EPI_FC, false);
    else { set_error("igemm fwd: unsupported epilogue %d", epi); return FRX_ERR_ARG; }
  }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
int launch_igemm_stem(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi) {
  if (epi == EPI_STATS) FRX_IGEMM_DT64(MODE_STEM, 0, EPI_STATS, false);
  else FRX_IGEMM_DT64(MODE_STEM, 0, EPI_PLAIN, false);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
}  // namespace frx

FRX_DBG_EXPORT(frx_debug_times_fwd)
