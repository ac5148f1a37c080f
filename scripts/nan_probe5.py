"""Diagnostic: unsynchronised replays, then inspect the non-finite conv outputs (where, how many, which values)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256; lr = 0.02
eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(1234)
batches = [((torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(), torch.randint(0, 10575, (N,), generator=g).cuda()) for _ in range(4)]
images = torch.empty_like(batches[0][0]); labels = torch.empty_like(batches[0][1])
eng.net.lr_dev.fill_(lr)
def feed(i):
    images.copy_(batches[i % 4][0]); labels.copy_(batches[i % 4][1])
feed(0)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eng.train_step(images, labels)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = eng.train_step(images, labels)
net = eng.net
for i in range(4):
    feed(i); gr.replay()
torch.cuda.synchronize()
print("loss", out["loss"].item())
for c in net.convs[:12]:
    y = c.y.float()
    nf = ~torch.isfinite(y)
    n = int(nf.sum())
    msg = f"{c.name:18s} shape {tuple(c.y.shape)} nonfinite {n}"
    if n:
        idx = nf.nonzero()
        msg += f" first {idx[0].tolist()} last {idx[-1].tolist()} nan {int(torch.isnan(y).sum())} +inf {int((y == float('inf')).sum())} -inf {int((y == float('-inf')).sum())}"
        ch = idx[:, 3].unique()
        msg += f" channels {ch[:8].tolist()}.. ({len(ch)})  pixels {len(idx[:, :3].unique(dim=0))}"
    print(msg, "| absmax finite", y[~nf].abs().max().item())
