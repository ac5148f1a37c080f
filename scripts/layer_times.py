"""Diagnostic: per-launch time of every GEMM-class kernel in one training step (HIP events, eager)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops, _lib
if os.environ.get("FRX_LIB"):          # A/B builds
    _lib.load_library(os.path.join(ROOT, os.environ["FRX_LIB"]))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
if os.environ.get("FRX_LT_BRANCH", "0") != "1":
    eng.net.branch_stream = None       # a launch is timed alone (the step's side stream would charge it for its neighbour)
g = torch.Generator().manual_seed(0)
images = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(); labels = torch.randint(0, 10575, (N,), generator=g).cuda()
# wrap to capture descriptors
descs = []
cur = [None]
orig = ops._timed
def timed(label, flops, t, fn, nbytes=0):
    if ops.PROFILER is not None:
        descs.append(cur[0])            # one entry per profiled launch (None: the grouped weight-gradient launch)
    cur[0] = None
    return orig(label, flops, t, fn, nbytes)
ops._timed = timed
for name in ("conv_fwd", "conv_fwd_tot", "conv_fwd_keep", "conv_fwd_merge", "conv_dgrad", "conv_wgrad", "conv_dgrad_bn", "conv_wgrad_bn"):
    f = getattr(ops, name)
    def mk(f, name):
        def w(d, *a, **k):
            cur[0] = (name, d); return f(d, *a, **k)
        return w
    setattr(ops, name, mk(f, name))
eng.train_step(images, labels, 0.1); eng.train_step(images, labels, 0.1)
torch.cuda.synchronize()
REP = 5
acc = None
for _ in range(REP):
    descs.clear(); ops.PROFILER = []
    eng.train_step(images, labels, 0.1)
    torch.cuda.synchronize()
    ts = [r[2].elapsed_time(r[3]) * 1e3 for r in ops.PROFILER]
    acc = ts if acc is None else [min(a, b) for a, b in zip(acc, ts)]
rec = ops.PROFILER; ops.PROFILER = None
tot = {}
print(f"{'op':13s} {'Ci':>5s} {'Co':>5s} k s {'Hi':>3s} {'us':>8s} {'TF/s':>7s} {'GB/s':>7s}  kernel class")
assert len(descs) == len(rec), (len(descs), len(rec))
for nd, (label, flops, _, _, nb), us in zip(descs, rec, acc):
    if nd is None:
        print(f"{'wgrad_group':13s} {'':5s} {'':5s}     {'':3s} {us:8.1f} {flops/us/1e6:7.1f} {nb/us/1e3:7.0f}  {label}")
        tot['wgrad_group'] = tot.get('wgrad_group', 0) + us
        continue
    name, d = nd
    M_out = d.N * d.Ho * d.Wo; M_in = d.N * d.Hi * d.Wi
    byt = nb or 2 * (M_in * (4 if d.stem else d.Ci) + M_out * d.Co + d.Co * d.R * d.S * d.Ci)
    print(f"{name:13s} {d.Ci:5d} {d.Co:5d} {d.R} {d.stride} {d.Hi:3d} {us:8.1f} {flops/us/1e6:7.1f} {byt/us/1e3:7.0f}  {label}")
    tot[name] = tot.get(name, 0) + us
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms")
