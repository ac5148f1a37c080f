"""Diagnostic for a box-dependent intermittent NaN: quick check, and if the box shows it, localise."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256; lr = 0.02; STEPS = 24
p = torch.cuda.get_device_properties(0); print("uuid", getattr(p, "uuid", None), flush=True)
g = torch.Generator().manual_seed(1234)
batches = [((torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(), torch.randint(0, 10575, (N,), generator=g).cuda()) for _ in range(4)]
def make(grouped=True):
    os.environ["FRX_WGRAD_GROUPED"] = "1" if grouped else "0"
    eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
    images = torch.empty_like(batches[0][0]); labels = torch.empty_like(batches[0][1])
    eng.net.lr_dev.fill_(lr)
    return eng, images, labels
FIXED_CAPTURE = False
def capture(eng, images, labels):
    side = torch.cuda.Stream()
    if FIXED_CAPTURE:
        side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.train_step(images, labels)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = eng.train_step(images, labels)
    return gr, out
def run(mode, grouped=True, watch_fn=None):
    eng, images, labels = make(grouped)
    def feed(i):
        images.copy_(batches[i % 4][0]); labels.copy_(batches[i % 4][1])
    feed(0)
    gr = out = None
    if mode.startswith("graph"): gr, out = capture(eng, images, labels)
    watch = watch_fn(eng) if watch_fn else []
    rec = torch.zeros(STEPS, max(1, len(watch)), 2, device="cuda:0")
    trace = torch.zeros(STEPS, device="cuda:0")
    for i in range(STEPS):
        feed(i)
        if gr is not None: gr.replay()
        else: out = eng.train_step(images, labels)
        trace[i].copy_(out["loss"].reshape(()))
        for k, (_, t) in enumerate(watch):
            tf = t.float() if t.dtype != torch.float32 else t
            rec[i, k, 0] = torch.isfinite(tf).all().float()
            rec[i, k, 1] = torch.nan_to_num(tf, nan=0.0, posinf=0.0, neginf=0.0).abs().max()
        if mode.endswith("_sync"): torch.cuda.synchronize()
    torch.cuda.synchronize()
    t = trace.tolist()
    bad = [i for i, v in enumerate(t) if v != v]
    return (bad[0] if bad else None), [round(v, 1) for v in t[:6]], rec.cpu(), watch
quick = [run("graph_nosync")[:2] for _ in range(3)]
print("quick check (graph, no sync):", quick, flush=True)
if all(q[0] is None for q in quick):
    print("box looks clean"); sys.exit(0)
print("=== BAD BOX: localising ===", flush=True)
FIXED_CAPTURE = True
print("graph_nosync with side.wait_stream(current) before the warm-up:", [run("graph_nosync")[:2] for _ in range(6)], flush=True)
FIXED_CAPTURE = False
print("graph_nosync as before:", [run("graph_nosync")[:2] for _ in range(3)], flush=True)
for mode in ("graph_sync", "eager_nosync", "eager_sync", "graph_nosync"):
    for grouped in (True, False):
        print(f"{mode:13s} grouped={int(grouped)}:", [run(mode, grouped)[:2] for _ in range(3)], flush=True)
def watch_fn(eng):
    net = eng.net
    w = [("xin", net.xin)]
    for c in net.convs:
        w += [(c.name + ".wk", c.wk), (c.name + ".y", c.y), (c.name + ".scale", net._bn(net.bn_scale, c)), (c.name + ".invstd", net._bn(net.bn_invstd, c))]
    for bi, b in enumerate(net.blocks): w.append((f"b{bi}.out", b.out))
    w += [("pooled", net.pooled), ("fc_wk", net.fc_wk), ("fc_b", net.fc_b()), ("feats", net.feats), ("dfeat", eng.dfeat)]
    for bi in range(len(net.blocks) - 1, -1, -1):
        b = net.blocks[bi]
        w += [(f"b{bi}.dz3", b.dz3), (f"b{bi}.dy2", b.dy2)] + [(f"b{bi}.dy[{k}]", v) for k, v in b.dyc.items()]
    w += [("g_pool", net.g_pool), ("dy_stem", net.dy_stem), ("grads", net.grads), ("params", net.params)]
    return w
for rep in range(3):
    first, head, rec, watch = run("graph_nosync", True, watch_fn)
    print(f"watched run {rep}: first NaN loss step {first}, losses {head}")
    fin = rec[:, :, 0] > 0.5
    steps_bad = (~fin).any(1).nonzero().flatten().tolist()
    if steps_bad:
        i = steps_bad[0]
        names = [watch[k][0] for k in range(len(watch)) if not fin[i, k]]
        print(f"   first step with a non-finite tensor: {i}; {len(names)} tensors; first ones: {names[:10]}")
        if i > 0:
            big = [(watch[k][0], rec[i - 1, k, 1].item(), rec[max(i - 2, 0), k, 1].item()) for k in range(len(watch))]
            big.sort(key=lambda x: -x[1])
            print("   largest |x| one step before:", [(n, f"{a:.3g}", f"{b_:.3g}") for n, a, b_ in big[:8]])
    sys.stdout.flush()
