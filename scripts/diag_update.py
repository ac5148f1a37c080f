"""Diagnostic: after one SGD step, compare every engine parameter with the oracle's."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch, torch.nn.functional as F
from oracle import heads as H
from oracle.resnet50 import FaceNet, make_sgd
from frx import engine as E, ops
N, C, lr = 8, 100, 2e-4
torch.manual_seed(1)
ref = FaceNet(H.ARC, C)
eng = E.FaceEngine("arcface", C, N, dtype=ops.F32, device="cuda:0")
eng.net.load_state_dict(ref.backbone.state_dict()); eng.head_w().copy_(ref.head.weight.detach().cuda())
w0 = {k: v.clone() for k, v in ref.backbone.state_dict().items()}
opt = make_sgd(ref, lr)
g = torch.Generator().manual_seed(1234)
for step in range(2):
    images = torch.rand(N, 3, 112, 112, generator=g) * 2 - 1
    labels = torch.randint(0, C, (N,), generator=g)
    eng.net.training = True
    eng.net.zero_grad(); out = eng.forward_loss(images.cuda(), labels.cuda(), want_logits=True); eng.backward(labels.cuda())
    ref.train(); (cs, lg), f = ref(images, labels); loss = F.cross_entropy(lg, labels); opt.zero_grad(); loss.backward()
    fe = F.normalize(out["feats"].cpu(), dim=1); fr = F.normalize(f.detach(), dim=1)
    print("step", step, "emb diff", (fe - fr).abs().max().item(), "loss", out["loss"].item(), loss.item())
    eng.net.sgd_step(lr); opt.step()
    sd = eng.net.state_dict(); rsd = ref.backbone.state_dict()
    bad = []
    for k in rsd:
        if "num_batches" in k: continue
        a, b = sd[k].cpu().float(), rsd[k].float()
        d = (a - b).norm().item() / (b.norm().item() + 1e-12)
        bad.append((d, k))
    bad.sort(reverse=True)
    print(" worst param rel diffs:", [(f"{d:.2e}", k) for d, k in bad[:6]])
    print(" head diff", ((eng.head_w().cpu() - ref.head.weight.detach()).norm() / ref.head.weight.norm()).item())
