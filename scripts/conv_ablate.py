"""Ablation: one conv shape, with/without BN prologue and stats epilogue (HIP-event timing)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
DEV = "cuda:0"
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (64, 64, 3, 1, 28), (128, 512, 1, 1, 14), (256, 1024, 1, 1, 7), (512, 512, 3, 1, 4), (256, 256, 3, 1, 7)]
N = 256
for Ci, Co, k, st, Hi in shapes:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, st, k // 2)
    x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16(); w = (torch.randn(Co, k, k, Ci, device=DEV) * 0.05).bfloat16()
    y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
    sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
    part = torch.empty(ops.conv_stat_rows(d), 2, Co, device=DEV)
    wt = w.permute(3, 1, 2, 0).contiguous(); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros(Co, k, k, Ci, device=DEV)
    t = {
        "full": bench(lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)),
        "nostat": bench(lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True)),
        "nopro": bench(lambda: ops.conv_fwd(d, x, w, y, stat_partial=part)),
        "plain": bench(lambda: ops.conv_fwd(d, x, w, y)),
        "dgrad": bench(lambda: ops.conv_dgrad(d, dy, wt, dx)),
        "dgrad+add": bench(lambda: ops.conv_dgrad(d, dy, wt, dx, addend=x)),
        "wgrad": bench(lambda: ops.conv_wgrad(d, x, dy, dw, in_scale=sc, in_shift=sh, in_relu=True)),
        "wgrad_nopro": bench(lambda: ops.conv_wgrad(d, x, dy, dw)),
    }
    fl = ops.conv_flops(d); byt = 2 * (x.numel() + y.numel() + w.numel())
    print(f"{Ci:4d}->{Co:4d} k{k} s{st} H{Hi:3d} | " + " ".join(f"{n}={v:6.1f}us" for n, v in t.items()) + f" | plain {fl/t['plain']/1e6:6.1f} TF {byt/t['plain']/1e3:6.0f} GB/s")
