"""What would the grouped weight-gradient lists cost if the 3x3 layers read a materialised, already normalised input (no
BN+ReLU prologue per staged tap)?  Timing only: the lists are re-planned with the prologue dropped for k == 3 (wrong dW)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256
e = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(1)
x = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(); y = torch.randint(0, 10575, (N,), generator=g).cuda()
for _ in range(2):
    e.train_step(x, y, 0.01)
net = e.net
def time_groups(tag):
    torch.cuda.synchronize()
    out = []
    for w in range(3):
        best = 1e9
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); net._run_wgrad_group(w); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        out.append(best)
    print(f"{tag:40s} lists (upper, layer2, layer1+stem): " + "  ".join(f"{t:7.1f} us" for t in out), flush=True)
time_groups("as planned")
orig = net._wgrad_job
def nopro(c, x, dy, x_bn=None, pro_y=None, pro_coef=None):
    return orig(c, x, dy, None if c.k == 3 else x_bn, pro_y, pro_coef)
net._wgrad_job = nopro
net._plan_wgrad_groups(); torch.cuda.synchronize()
time_groups("3x3 inputs without prologue")
def nopro_all(c, x, dy, x_bn=None, pro_y=None, pro_coef=None):
    return orig(c, x, dy, None, pro_y, pro_coef)
net._wgrad_job = nopro_all
net._plan_wgrad_groups(); torch.cuda.synchronize()
time_groups("no input prologue anywhere")
# how much of each list is the 3x3 layers' work?  (lists re-planned WITHOUT them)
import types
src_plan = E.ResNet50Engine._plan_wgrad_groups if hasattr(E, "ResNet50Engine") else None
net._wgrad_job = orig
real_plan = ops.wgrad_group_plan
def plan_without_3x3(dtype, jobs):
    return real_plan(dtype, [j for j in jobs if not (j["d"].R == 3 and not j["d"].stem)])
ops.wgrad_group_plan = plan_without_3x3
net._plan_wgrad_groups(); torch.cuda.synchronize()
time_groups("without the 3x3 layers' jobs")
ops.wgrad_group_plan = real_plan
