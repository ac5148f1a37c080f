"""Diagnostic: wgrad time vs K-chunks per block at a fixed grid (fixed overhead vs per-chunk cost)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
DEV = "cuda:0"
def run(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
os.environ["FRX_WGRAD_MINCHUNKS"] = "1"
for (Ci, Co, k, Hi, bn) in [(256, 1024, 1, 7, True), (256, 1024, 1, 7, False), (256, 256, 3, 7, False)]:
    for blocks in (128, 256, 512, 1024):
        os.environ["FRX_WGRAD_BLOCKS"] = str(blocks)
        row = []
        for N in (16, 32, 64, 128, 256, 512, 1024):
            d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, 1, k // 2)
            x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16()
            dy = torch.randn(N, d.Ho, d.Wo, Co, device=DEV).bfloat16(); y2 = torch.randn_like(dy)
            sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
            coef = torch.randn(3, Co, device=DEV)
            dw = torch.zeros(Co, k, k, Ci, device=DEV)
            if bn: fn = lambda: ops.conv_wgrad_bn(d, x, dy, y2, coef, dw, in_scale=sc, in_shift=sh, in_relu=True)
            else: fn = lambda: ops.conv_wgrad(d, x, dy, dw)
            tiles = (Co // 128) * (Ci // 128) * k * k
            splits = max(1, -(-blocks // tiles)); 
            if splits >= 8: splits = splits // 8 * 8
            nch = -(-N * Hi * Hi // 32)
            cps = -(-nch // splits)
            row.append(f"{cps:4d}ch {run(fn):6.1f}us")
        print(f"{Ci}->{Co} k{k} bn={int(bn)} blocks~{blocks:5d}:", " | ".join(row), flush=True)
