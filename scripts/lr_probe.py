import sys, os
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N=256
for lr in (0.02, 0.01, 0.005, 0.002):
    eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
    g = torch.Generator().manual_seed(0)
    images = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(); labels = torch.randint(0, 10575, (N,), generator=g).cuda()
    ls=[]
    for i in range(60):
        o = eng.train_step(images, labels, lr)
        if i % 6 == 0: ls.append(round(o["loss"].item(),2))
    print(lr, ls, flush=True)
