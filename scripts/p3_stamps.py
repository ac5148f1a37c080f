"""Diagnostic (needs a -DFRX_DBG_TIMES build): per-block phase timestamps of the patch-mode 3x3 launches of k_igemm."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch, numpy as np
from frx import ops, _lib
from igemm_stamps import stamps, report, timeit
DEV = "cuda:0"; N = 256
for (C_, Hi) in [(64, 28), (128, 14), (256, 7), (512, 4)]:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, C_, C_, 3, 3, 1, 1)
    x = torch.randn(N, Hi, Hi, C_, device=DEV).bfloat16(); w = (torch.randn(C_, 3, 3, C_, device=DEV) * 0.05).bfloat16()
    y = torch.empty(N, Hi, Hi, C_, device=DEV, dtype=torch.bfloat16)
    sc = torch.rand(C_, device=DEV) + 0.5; sh = torch.randn(C_, device=DEV) * 0.1
    tot = torch.zeros(8, 2, C_, device=DEV)
    fn = lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True)
    us = timeit(fn)
    M = N * Hi * Hi
    nb = ((-(-M // 128) + 7) // 8 * 8) * max(1, C_ // 128)
    report(f"p3 fwd {C_}->{C_} H{Hi}", stamps("frx_debug_times_p3", nb), us)
    wt = w.permute(3, 1, 2, 0).contiguous(); dz = torch.randn_like(y); dx = torch.empty_like(x)
    coef = torch.randn(3, C_, device=DEV); ey = torch.randn_like(x)
    esc = torch.rand(C_, device=DEV) + 0.5; esh = torch.randn(C_, device=DEV) * 0.1
    emu = torch.randn(C_, device=DEV); eis = torch.rand(C_, device=DEV) + 0.5
    fn = lambda: ops.conv_dgrad_bn(d, dz, wt, dx, pro_y=y, pro_coef=coef, epi_y=ey, epi_scale=esc, epi_shift=esh,
                                   epi_mean=emu, epi_invstd=eis, epi_totals=tot, epi_replicas=8)
    us = timeit(fn)
    report(f"p3 dgrad {C_}->{C_} H{Hi}", stamps("frx_debug_times_p3", nb), us)
