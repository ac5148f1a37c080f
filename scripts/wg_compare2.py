"""Diagnostic: is it the grouped schedule or the FIRST engine of the process that differs?  Also compares the forward."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N, C = 256, 10575
def eng(grouped):
    os.environ["FRX_WGRAD_GROUPED"] = "1" if grouped else "0"
    return E.FaceEngine("arcface", C, N, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(2)
x = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(); y = torch.randint(0, C, (N,), generator=g).cuda()
order = [False, True, False, True, True]
res = []
for grouped in order:
    e = eng(grouped)
    e.net.zero_grad(); o = e.forward_loss(x, y); feats = o["feats"].clone(); loss = o["loss"].item(); e.backward(y)
    torch.cuda.synchronize()
    res.append((grouped, feats, loss, e.dfeat.clone(), e.net.grads.clone(), e.net.blocks[-1].dz3.clone().float(), e.net.params.clone()))
ref = res[2]
for i, r in enumerate(res):
    d = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
    print(f"run {i} grouped={int(r[0])}: loss {r[2]:.6f} | vs run 2: params {d(r[6], ref[6]):.1e} feats {d(r[1], ref[1]):.1e} dfeat {d(r[3], ref[3]):.1e} "
          f"top dz3 {d(r[5], ref[5]):.1e} grads {d(r[4], ref[4]):.1e}")
