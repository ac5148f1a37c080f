"""Diagnostic: the bench's training loop (4 rotating random batches, lr 0.02) with the loss read every step."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256; lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
for trial in range(3):
    eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
    g = torch.Generator().manual_seed(1234)
    batches = [((torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(), torch.randint(0, 10575, (N,), generator=g).cuda()) for _ in range(4)]
    ls = []
    for i in range(45):
        o = eng.train_step(*batches[i % 4], lr)
        l = o["loss"].item()
        ls.append(round(l, 1))
        if l != l:
            p = eng.net.params
            print("  NaN at step", i, "params finite:", bool(torch.isfinite(p).all()), "grads finite:", bool(torch.isfinite(eng.net.grads).all()),
                  "max|p|", p[torch.isfinite(p)].abs().max().item())
            # which BN stats are broken
            bad = [c.name for c in eng.net.convs if not torch.isfinite(eng.net._bn(eng.net.bn_invstd, c)).all()]
            print("  convs with non-finite invstd:", bad[:6])
            bady = [c.name for c in eng.net.convs if not torch.isfinite(c.y.float()).all()]
            print("  convs with non-finite output:", bady[:6])
            break
    print("trial", trial, ls, flush=True)
