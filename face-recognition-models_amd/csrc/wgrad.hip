// k_wgrad instantiations.
#include "conv_launch.h"
#include <stdlib.h>
namespace frx {
#define FRX_WG(T_, BT_, WM_, PRO_, YP_) hipLaunchKernelGGL((k_wgrad<T_, BT_, WM_, PRO_, YP_>), dim3(grid), dim3(256), 0, st, a)
#define FRX_WG_Y(T_, BT_, WM_, PRO_) do { if (ypro) FRX_WG(T_, BT_, WM_, PRO_, true); else FRX_WG(T_, BT_, WM_, PRO_, false); } while (0)
#define FRX_WG_MODE(T_, BT_)                                                                          \
  do {                                                                                                \
    if (wmode == WG_STEM) FRX_WG(T_, BT_, WG_STEM, false, false);                                     \
    else if (wmode == WG_POINTWISE) { if (pro) FRX_WG_Y(T_, BT_, WG_POINTWISE, true); else FRX_WG_Y(T_, BT_, WG_POINTWISE, false); } \
    else { if (pro) FRX_WG_Y(T_, BT_, WG_GENERAL, true); else FRX_WG_Y(T_, BT_, WG_GENERAL, false); }                               \
  } while (0)
int launch_wgrad(hipStream_t st, const WgradArgs& a, int dtype, int bt, int wmode, bool pro, bool ypro, int grid) {
  if (dtype == FRX_BF16) {
    if (bt == 64) FRX_WG_MODE(bf16_t, 64); else FRX_WG_MODE(bf16_t, 128);
  } else {
    if (bt == 64) FRX_WG_MODE(float, 64); else FRX_WG_MODE(float, 128);
  }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
// persistent blocks: two per CU (the register budget of the widest variant) -- four for a list of 64 x 64-tile layers only --
// fewer when the list is short
int launch_wgrad_grouped(hipStream_t st, int dtype, const WgradArgs* layers, const WgradItem* items, int nitems, bool small_tiles,
                         int* draw_counters) {
  int grid = small_tiles ? 1024 : 512;
  if (grid > nitems) grid = nitems;
  grid = grid / 8 * 8;
  if (grid < 8) grid = 8;
  if (dtype == FRX_BF16) {
    if (small_tiles) hipLaunchKernelGGL((k_wgrad_grouped<bf16_t, true>), dim3(grid), dim3(256), 0, st, layers, items, nitems, draw_counters);
    else hipLaunchKernelGGL((k_wgrad_grouped<bf16_t, false>), dim3(grid), dim3(256), 0, st, layers, items, nitems, draw_counters);
  } else {
    if (small_tiles) hipLaunchKernelGGL((k_wgrad_grouped<float, true>), dim3(grid), dim3(256), 0, st, layers, items, nitems, draw_counters);
    else hipLaunchKernelGGL((k_wgrad_grouped<float, false>), dim3(grid), dim3(256), 0, st, layers, items, nitems, draw_counters);
  }
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}
}  // namespace frx

FRX_DBG_EXPORT(frx_debug_times_wgrad)
