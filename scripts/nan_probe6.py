"""Diagnostic: same box, same process -- which execution mode produces the intermittent NaN?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256; lr = 0.02; STEPS = 30
p = torch.cuda.get_device_properties(0); print("uuid", getattr(p, "uuid", None), flush=True)
g = torch.Generator().manual_seed(1234)
batches = [((torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(), torch.randint(0, 10575, (N,), generator=g).cuda()) for _ in range(4)]
def run(mode, grouped):
    os.environ["FRX_WGRAD_GROUPED"] = "1" if grouped else "0"
    eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
    images = torch.empty_like(batches[0][0]); labels = torch.empty_like(batches[0][1])
    eng.net.lr_dev.fill_(lr)
    def feed(i):
        images.copy_(batches[i % 4][0]); labels.copy_(batches[i % 4][1])
    feed(0)
    gr = None
    if mode.startswith("graph"):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            eng.train_step(images, labels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = eng.train_step(images, labels)
    trace = torch.zeros(STEPS, device="cuda:0")
    for i in range(STEPS):
        feed(i)
        if gr is not None: gr.replay()
        else: out = eng.train_step(images, labels)
        trace[i].copy_(out["loss"].reshape(()))
        if mode.endswith("sync"): torch.cuda.synchronize()
    torch.cuda.synchronize()
    t = trace.tolist()
    bad = [i for i, v in enumerate(t) if v != v]
    return (bad[0] if bad else None), round(t[1], 2)
for mode in ("graph_nosync", "graph_sync", "eager_nosync", "graph_nosync"):
    for grouped in (True, False):
        res = [run(mode, grouped) for _ in range(3)]
        print(f"{mode:13s} grouped={int(grouped)}: first NaN step / loss[1]:", res, flush=True)
