"""Diagnostic: bf16 vs fp32 engine forward, per-layer relative difference of raw conv outputs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
e32 = E.FaceEngine("cosface", 100, N, dtype=ops.F32, device="cuda:0", seed=0)
e16 = E.FaceEngine("cosface", 100, N, dtype=ops.BF16, device="cuda:0", seed=0)
e16.net.load_state_dict(e32.net.state_dict()); e16.head_w().copy_(e32.head_w())
g = torch.Generator().manual_seed(7)
x = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(); y = torch.randint(0, 100, (N,), generator=g).cuda()
o32 = e32.forward_loss(x, y); o16 = e16.forward_loss(x, y)
print("loss", o32["loss"].item(), o16["loss"].item())
for c32, c16 in zip(e32.net.convs, e16.net.convs):
    a, b = c32.y.float(), c16.y.float()
    rel = (a - b).norm().item() / (a.norm().item() + 1e-12)
    sc = (e32.net._bn(e32.net.bn_scale, c32) - e16.net._bn(e16.net.bn_scale, c16)).abs().max().item()
    print(f"{c32.name:26s} rel {rel:.3e}  dscale {sc:.3e}  nan16 {torch.isnan(b).any().item()}")
for b32, b16 in zip(e32.net.blocks, e16.net.blocks):
    a, b = b32.out.float(), b16.out.float()
    print("block out rel", (a - b).norm().item() / a.norm().item())
