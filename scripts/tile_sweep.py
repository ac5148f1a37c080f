"""Diagnostic: forward / fused-dgrad time of each ResNet-50 conv shape under each igemm tile (FRX_IGEMM_TILE)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
DEV = "cuda:0"; N = 256
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
SH = [(64, 64, 1, 1, 28), (64, 64, 3, 1, 28), (64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (256, 128, 1, 1, 28), (128, 128, 3, 2, 28),
      (128, 512, 1, 1, 14), (512, 128, 1, 1, 14), (128, 128, 3, 1, 14), (512, 256, 1, 1, 14), (256, 256, 3, 2, 14), (256, 1024, 1, 1, 7),
      (1024, 256, 1, 1, 7), (256, 256, 3, 1, 7), (1024, 512, 1, 1, 7), (512, 512, 3, 2, 7), (512, 2048, 1, 1, 4), (2048, 512, 1, 1, 4),
      (512, 512, 3, 1, 4), (256, 512, 1, 2, 28), (512, 1024, 1, 2, 14), (1024, 2048, 1, 2, 7)]
TILES = ["default", "128x128", "128x64", "64x64"]
print("shape".ljust(22), *[("fwd " + t).rjust(12) for t in TILES], *[("dgrad " + t).rjust(13) for t in TILES])
for (Ci, Co, k, st, Hi) in SH:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, st, k // 2)
    x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16(); w = (torch.randn(Co, k, k, Ci, device=DEV) * 0.05).bfloat16()
    y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
    sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
    wt = w.permute(3, 1, 2, 0).contiguous(); dz = torch.randn_like(y); dx = torch.empty_like(x)
    coef = torch.randn(3, Co, device=DEV); ey = torch.randn_like(x)
    emu = torch.randn(Ci, device=DEV); eis = torch.rand(Ci, device=DEV) + 0.5
    rf, rd = [], []
    for t in TILES:
        os.environ.pop("FRX_IGEMM_TILE", None)
        if t != "default": os.environ["FRX_IGEMM_TILE"] = t
        part = torch.empty(ops.conv_stat_rows(d), 2, Co, device=DEV)
        rf.append(timeit(lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)))
        ep = torch.empty(ops.conv_dgrad_stat_rows(d), 2, Ci, device=DEV)
        if k == 1:
            fn = lambda: ops.conv_dgrad_bn(d, dz, wt, dx, pro_y=y, pro_coef=coef, epi_y=ey, epi_scale=sc, epi_shift=sh,
                                           epi_mean=emu, epi_invstd=eis, epi_partial=ep)
        else:
            fn = lambda: ops.conv_dgrad_bn(d, dz, wt, dx, epi_y=ey, epi_scale=sc, epi_shift=sh, epi_mean=emu, epi_invstd=eis, epi_partial=ep)
        rd.append(timeit(fn))
    print(f"{Ci}->{Co} k{k} s{st} H{Hi}".ljust(22), *[f"{v:12.1f}" for v in rf], *[f"{v:13.1f}" for v in rd], flush=True)
