"""Measurements quoted in DESIGN.md besides the headline bench:
  (1) PCIe-inclusive training rate: each step also uploads a fresh uint8 NHWC batch from pinned host memory;
  (2) BASELINE config 5: 6000-pair LFW-style verification (12 000 images embedded once at B=512, pair cosine,
      10-fold threshold protocol) on one GPU, and the same arithmetic on the CPU oracle for a sample."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import numpy as np, torch
from frx import engine as E, ops
dev = torch.device("cuda:0")
out = {}
# ---- (1)
N, C = 256, 10575
eng = E.FaceEngine("arcface", C, N, dtype=ops.BF16, device=dev, seed=0)
eng.net.lr_dev.fill_(0.02)
g = torch.Generator().manual_seed(0)
host = [torch.randint(0, 256, (N, 112, 112, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(4)]
lab = torch.randint(0, C, (N,), generator=g).to(dev)
dimg = torch.empty(N, 112, 112, 3, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    eng.train_step(dimg, lab)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    eng.train_step(dimg, lab)
for mode in ("resident", "pcie"):
    for i in range(3):
        if mode == "pcie": dimg.copy_(host[i % 4], non_blocking=True)
        graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 30
    for i in range(K):
        if mode == "pcie": dimg.copy_(host[i % 4], non_blocking=True)
        graph.replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[f"train_u8_{mode}_img_per_s"] = round(N * K / dt, 1)
del graph, eng
# ---- (2)
P, B = 6000, 512
eng = E.FaceEngine("arcface", 100, B, dtype=ops.BF16, device=dev, seed=1)
imgs = torch.randint(0, 256, (2 * P, 112, 112, 3), dtype=torch.uint8, generator=g)     # 12 000 synthetic faces
same = np.r_[np.ones(P // 2), np.zeros(P // 2)].astype(np.int64); np.random.RandomState(0).shuffle(same)
sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
from utils import model_utils as MU
torch.cuda.synchronize(); t0 = time.perf_counter()
emb = []
pinned = imgs.pin_memory()
for lo in range(0, 2 * P, B):
    chunk = pinned[lo:lo + B]
    if chunk.shape[0] < B:                       # last ragged batch: pad to the planned batch
        pad = torch.zeros(B, 112, 112, 3, dtype=torch.uint8); pad[:chunk.shape[0]] = chunk; chunk = pad.pin_memory()
    emb.append(eng.embed(chunk.to(dev, non_blocking=True)).clone())
emb = torch.cat(emb)[:2 * P]
cos = ops.pair_cosine(emb[:P].contiguous(), emb[P:].contiguous())
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    res = MU.kfold_from_similarities(cos, same, 10)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["lfw6000_end_to_end_s"] = round(dt, 3)
out["lfw6000_embeds_per_s"] = round(2 * P / dt, 1)
out["lfw6000_result"] = [round(v, 4) for v in res]
# oracle arithmetic on the same similarities (accuracy parity of the protocol itself)
from oracle import verify as OV
(ma, sa, mu, su), _, _ = OV.cross_validate_kfold(cos.cpu().numpy(), same, 10)
out["lfw6000_oracle_protocol_acc"] = round(ma, 4)
# ---- (3) the drop-in loop a user of arcface.py gets: utils.model_utils.train_model on ArcFaceNet with FusedSGD (hipGraph replay
# from the second step on, copy-stream prefetch), fed uint8 HWC batches and, separately, the reference's fp32 CHW batches
del eng
import types, torch.nn as nn
from utils import criterion as UC
for mode in ("uint8", "fp32"):
    torch.manual_seed(0)
    model = UC.ArcFaceNet(num_classes=C, backbone="resnet50").to(dev)
    opt = MU.make_optimizer(model, 0.005)
    if mode == "uint8":
        data = [(host[i % 4], lab.cpu()) for i in range(44)]
    else:
        f = [((h.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5).contiguous().pin_memory() for h in host]
        data = [(f[i % 4], lab.cpu()) for i in range(44)]
    args = types.SimpleNamespace(lambda_g=0.0, print_freq=1000)
    crit = nn.CrossEntropyLoss().to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        MU.train_model(model, data[:4], crit, opt, MU.GradScaler(enabled=False), dev, 1, 1, args)      # engine build, eager step, capture
        torch.cuda.synchronize(); t0 = time.perf_counter()
        MU.train_model(model, data[4:], crit, opt, MU.GradScaler(enabled=False), dev, 1, 1, args)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[f"train_model_dropin_{mode}_img_per_s"] = round(N * 40 / dt, 1)
    del model, opt
print(json.dumps(out))
