"""Diagnostic: unsynchronised graph replays (as bench.py issues them) with on-device finiteness flags per step."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256; lr = 0.005; STEPS = 40
p_ = torch.cuda.get_device_properties(0); print("uuid", getattr(p_, "uuid", None), flush=True)
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
    g = torch.Generator().manual_seed(1234)
    batches = [((torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(), torch.randint(0, 10575, (N,), generator=g).cuda()) for _ in range(4)]
    images = torch.empty_like(batches[0][0]); labels = torch.empty_like(batches[0][1])
    eng.net.lr_dev.fill_(lr)
    def feed(i):
        images.copy_(batches[i % 4][0]); labels.copy_(batches[i % 4][1])
    feed(0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.train_step(images, labels)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = eng.train_step(images, labels)
    net = eng.net
    watch = [("xin", net.xin), ("stem.y", net.stem.y), ("pool", net.pool_out)]
    for bi, b in enumerate(net.blocks):
        watch += [(f"b{bi}.c1y", b.conv1.y), (f"b{bi}.c2y", b.conv2.y), (f"b{bi}.c3y", b.conv3.y), (f"b{bi}.out", b.out)]
    watch += [("feats", net.feats), ("loss", out["loss"]), ("dfeat", eng.dfeat)]
    for bi in range(len(net.blocks) - 1, -1, -1):
        b = net.blocks[bi]
        watch += [(f"b{bi}.dz3", b.dz3), (f"b{bi}.dy2", b.dy2)] + [(f"b{bi}.dy[{k}]", v) for k, v in b.dyc.items()]
    watch += [("g_pool", net.g_pool), ("dy_stem", net.dy_stem), ("grads", net.grads), ("params", net.params),
              ("bn_mean", net.bn_mean), ("bn_invstd", net.bn_invstd), ("bn_scale", net.bn_scale), ("bn_shift", net.bn_shift)]
    flags = torch.ones(STEPS, len(watch), dtype=torch.bool, device="cuda:0")
    amax = torch.zeros(STEPS, len(watch), device="cuda:0")
    for i in range(STEPS):
        feed(i)
        gr.replay()
        for k, (_, t) in enumerate(watch):
            flags[i, k] = torch.isfinite(t).all()
            amax[i, k] = torch.nan_to_num(t.float(), nan=0.0, posinf=0.0, neginf=0.0).abs().max()
    torch.cuda.synchronize()
    f = flags.cpu()
    bad = (~f).any(1).nonzero().flatten().tolist()
    if bad:
        i = bad[0]
        names = [watch[k][0] for k in range(len(watch)) if not f[i, k]]
        last = [watch[k][0] for k in range(len(watch)) if not f[STEPS - 1, k]]
        print(f"trial {trial}: first non-finite at replay {i}: {len(names)} tensors, loss bad: {'loss' in names}, params bad: {'params' in names};"
              f" bad steps {bad[:10]}; at the last step {len(last)} bad, loss bad {'loss' in last}; final loss {out['loss'].item():.2f}", flush=True)
        print("    ", names[:6], "...", names[-8:])
        am = amax.cpu()
        for nm in ("grads", "params", "feats", "dfeat", "b15.out", "b0.dz3", "g_pool", "dy_stem", "bn_invstd", "bn_scale"):
            k = [w[0] for w in watch].index(nm)
            print(f"     |{nm}|max over the steps before: ", [f"{am[j, k].item():.3g}" for j in range(max(0, i - 4), i + 1)])
    else:
        print(f"trial {trial}: {STEPS} unsynchronised replays clean", flush=True)
