"""Diagnostic: forward conv with and without the BN prologue / statistics epilogue (what the fusion costs per shape)."""
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
SH = [(64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (64, 64, 3, 1, 28), (128, 512, 1, 1, 14), (512, 128, 1, 1, 14), (128, 128, 3, 1, 14),
      (256, 1024, 1, 1, 7), (1024, 256, 1, 1, 7), (256, 256, 3, 1, 7), (512, 2048, 1, 1, 4), (2048, 512, 1, 1, 4), (512, 512, 3, 1, 4)]
print("shape".ljust(20), "plain".rjust(8), "+stats".rjust(8), "+pro".rjust(8), "+both".rjust(8), "TF(both)".rjust(9))
for (Ci, Co, k, st, Hi) in SH:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, st, k // 2)
    x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16(); w = (torch.randn(Co, k, k, Ci, device=DEV) * 0.05).bfloat16()
    y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
    sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
    part = torch.empty(ops.conv_stat_rows(d), 2, Co, device=DEV)
    t = [timeit(lambda: ops.conv_fwd(d, x, w, y)),
         timeit(lambda: ops.conv_fwd(d, x, w, y, stat_partial=part)),
         timeit(lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True)),
         timeit(lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part))]
    fl = 2.0 * N * d.Ho * d.Wo * Ci * Co * k * k
    print(f"{Ci}->{Co} k{k} H{Hi}".ljust(20), *[f"{v:8.1f}" for v in t], f"{fl/t[3]/1e6:9.0f}", flush=True)
