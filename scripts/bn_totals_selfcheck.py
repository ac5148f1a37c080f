"""Debug aid: self-consistency of the replicated-totals forward inside ONE engine (no second trajectory to diverge from):
every layer's mean / invstd against torch reductions of that layer's own raw output, every block output against the merge
recomputed from the engine's own constants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
os.environ["FRX_BN_DETERMINISTIC"] = sys.argv[2] if len(sys.argv) > 2 else "0"
f = E.FaceEngine("arcface", 1000, N, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(5)
x = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(); y = torch.randint(0, 1000, (N,), generator=g).cuda()
f.net.training = True
f.forward_loss(x, y)
torch.cuda.synchronize()
n = f.net
print("fused_bn", n.fused_bn)
for c in n.convs:
    yy = c.y.float().reshape(-1, c.Co)
    m, v = yy.mean(0), yy.var(0, unbiased=False)
    em = (n._bn(n.bn_mean, c) - m).abs().max().item() / (m.abs().max().item() + 1e-6)
    ei = (n._bn(n.bn_invstd, c) - (v + 1e-5).rsqrt()).abs().max().item() / (v + 1e-5).rsqrt().abs().max().item()
    flag = "  <<<<" if max(em, ei) > 1e-3 else ""
    print(f"{c.name:24s} mean err {em:.2e} invstd err {ei:.2e}{flag}")
xin = n.pool_out
for bi, b in enumerate(n.blocks):
    s3, h3 = n._bn(n.bn_scale, b.conv3), n._bn(n.bn_shift, b.conv3)
    idn = xin.float() if b.down is None else b.down.y.float() * n._bn(n.bn_scale, b.down) + n._bn(n.bn_shift, b.down)
    ref = torch.relu(b.conv3.y.float() * s3 + h3 + idn)
    err = (b.out.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
    print(f"block {bi:2d} out err {err:.2e}" + ("  <<<<" if err > 2e-2 else ""))
    xin = b.out
