import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/face-recognition-models_amd")
import torch
from frx import engine as E, ops
eng = E.FaceEngine("arcface", 10575, 256, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(256, 3, 112, 112, generator=g) * 2 - 1).cuda(); y = torch.randint(0, 10575, (256,), generator=g).cuda()
eng.net.lr_dev.fill_(0.02)
for _ in range(2): eng.train_step(x, y)
torch.cuda.synchronize()
ops.PROFILER = []
eng.train_step(x, y)
torch.cuda.synchronize()
for label, fl, e0, e1, nb in ops.PROFILER:
    if "wgrad<bf16,64>" in label: print(label, round(e0.elapsed_time(e1) * 1e3, 1), "us", nb)
