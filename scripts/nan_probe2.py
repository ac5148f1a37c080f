"""Diagnostic: bench-style hipGraph replay loop with the loss read every step (hunting an intermittent NaN)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256; lr = 0.02
for trial in range(4):
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
    ls = []
    sync_each = trial % 2 == 0
    for i in range(40):
        feed(i)
        gr.replay()
        if sync_each:
            l = out["loss"].item(); ls.append(round(l, 1))
            if l != l:
                print("  NaN at replay", i); break
    torch.cuda.synchronize()
    l = out["loss"].item()
    p = eng.net.params
    print("trial", trial, "sync_each", sync_each, "final loss", l, "params finite", bool(torch.isfinite(p).all()), ls, flush=True)
