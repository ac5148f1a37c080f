// k_igemm instantiations: 1x1 forward with the residual merge of the block before as its prologue (PRO = 3).
// One tile per block (the persistent form of these variants spills: not instantiated, conv.hip keeps grid = tiles).
#include "conv_launch.h"
namespace frx {
#define FRX_MERGE_LAUNCH(T_, EPI_)                                                                              \
  do {                                                                                                          \
    if (c.kc == 128 && c.bm == 64 && c.bn == 128) FRX_IGEMM_K(T_, 64, 128, 1, 4, MODE_FWD, 3, EPI_, false, 128); \
    else if (c.bm == 128 && c.bn == 128) FRX_IGEMM_K(T_, 128, 128, 2, 4, MODE_FWD, 3, EPI_, false, 64);          \
    else if (c.bm == 128 && c.bn == 64) FRX_IGEMM_K(T_, 128, 64, 2, 2, MODE_FWD, 3, EPI_, false, 64);            \
    else FRX_IGEMM_K(T_, 64, 64, 2, 2, MODE_FWD, 3, EPI_, false, 64);                                            \
  } while (0)
int launch_igemm_fwd_merge(hipStream_t st, const ConvArgs& a, int dtype, TileCfg c, int grid, int epi) {
  if (grid != a.nvb) { set_error("igemm fwd (merge prologue): one tile per block"); return FRX_ERR_ARG; }
  if (epi == EPI_STATS) { if (dtype == FRX_BF16) FRX_MERGE_LAUNCH(bf16_t, EPI_STATS); else FRX_MERGE_LAUNCH(float, EPI_STATS); }
  else if (epi == EPI_PLAIN) { if (dtype == FRX_BF16) FRX_MERGE_LAUNCH(bf16_t, EPI_PLAIN); else FRX_MERGE_LAUNCH(float, EPI_PLAIN); }
  else { set_error("igemm fwd (merge prologue): unsupported epilogue %d", epi); return FRX_ERR_ARG; }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
}  // namespace frx

FRX_DBG_EXPORT(frx_debug_times_fwd_merge)
