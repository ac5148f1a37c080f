"""How far do two bf16 forward passes of the random-init network drift apart from a last-bit difference?  The DETERMINISTIC
engine (partial rows + finalize launches: bit-reproducible) is run on a batch and on the same batch with ONE input value
moved by one bf16 ulp; relative L2 difference of the embeddings and of layer outputs along the depth.  This is the
yardstick for comparing two runs of the replicated-totals path (whose BatchNorm sums differ in their last bits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
os.environ["FRX_BN_DETERMINISTIC"] = "1"
from frx import engine as E, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
e = E.FaceEngine("cosface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(11)
x = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda()
y = torch.randint(0, 10575, (N,), generator=g).cuda()
rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
e.net.training = True
f0 = e.forward_loss(x, y)["feats"].clone()
outs0 = [b.out.clone() for b in e.net.blocks]
f0b = e.forward_loss(x, y)["feats"].clone()
print("same input twice (deterministic engine): rel diff", rel(f0b, f0))
x2 = x.clone(); x2[0, 0, 56, 56] += 2.0 ** -8
f1 = e.forward_loss(x2, y)["feats"].clone()
print(f"one input value moved by 2^-8: embeddings rel L2 diff {rel(f1, f0):.3e}")
print("block outputs rel L2 diff by depth:", " ".join(f"{rel(b.out, o):.1e}" for b, o in zip(e.net.blocks, outs0)))
os.environ["FRX_BN_DETERMINISTIC"] = "0"
t = E.FaceEngine("cosface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
t.net.training = True
a = t.forward_loss(x, y)["feats"].clone(); outs1 = [b.out.clone() for b in t.net.blocks]
b_ = t.forward_loss(x, y)["feats"].clone()
print(f"replicated totals, same input twice: embeddings rel L2 diff {rel(b_, a):.3e}; vs the deterministic engine {rel(a, f0):.3e}")
print("block outputs rel L2 diff by depth (two runs of the totals engine):", " ".join(f"{rel(b.out, o):.1e}" for b, o in zip(t.net.blocks, outs1)))
