"""Diagnostic: what does one extra tiny kernel cost inside the replayed step (unprofiled)?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256
eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(0)
images = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(); labels = torch.randint(0, 10575, (N,), generator=g).cuda()
eng.net.lr_dev.fill_(0.005)
dummy = torch.zeros(64, device="cuda:0")
def step(extra):
    out = eng.train_step(images, labels)
    for _ in range(extra): dummy.add_(1.0)
    return out
for extra in (0, 100, 200, 400):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): step(extra)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): step(extra)
    for _ in range(5): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): gr.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"extra tiny kernels per step: {extra:4d}  ->  {dt*1e3:.3f} ms/step", flush=True)
