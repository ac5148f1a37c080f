"""Run ONE conv shape a few times (for rocprofv3 --pmc): python one_conv.py Ci Co k stride Hi [mode]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
Ci, Co, k, st, Hi = map(int, sys.argv[1:6]); mode = sys.argv[6] if len(sys.argv) > 6 else "fwd"
N = 256; DEV = "cuda:0"
d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, st, k // 2)
x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16(); w = (torch.randn(Co, k, k, Ci, device=DEV) * 0.05).bfloat16()
y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
part = torch.empty(ops.conv_stat_rows(d), 2, Co, device=DEV)
wt = w.permute(3, 1, 2, 0).contiguous(); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros(Co, k, k, Ci, device=DEV)
for _ in range(5):
    if mode == "fwd": ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)
    elif mode == "plain": ops.conv_fwd(d, x, w, y)
    elif mode == "dgrad": ops.conv_dgrad(d, dy, wt, dx)
    elif mode == "wgrad": ops.conv_wgrad(d, x, dy, dw, in_scale=sc, in_shift=sh, in_relu=True)
torch.cuda.synchronize()
