"""Head-only timing: forward (both phases) + backward of one margin head at a BASELINE size, HIP events around
K iterations; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
    python scripts/head_bench.py [kind=curricular] [N=128] [C=85000] [iters=20]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops, _lib
if os.environ.get("FRX_LIB"):          # A/B builds
    _lib.load_library(os.path.join(ROOT, os.environ["FRX_LIB"]))
kind = sys.argv[1] if len(sys.argv) > 1 else "curricular"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
C = int(sys.argv[3]) if len(sys.argv) > 3 else 85000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
DEV = torch.device("cuda:0")
kid = E.HEAD_KINDS[kind]
s_, m_ = E.HEAD_DEFAULTS[kid]
FLAGS = E.HEAD_FLAG_DEFAULTS.get(kid, 0)
ctx = ops.HeadContext(kid, N, 512, C, s_, m_, 0.01, device=DEV, p=E.HEAD_P_DEFAULTS.get(kid, ()), flags=FLAGS)
g = torch.Generator().manual_seed(0)
cd = kid in ops.W_CD_KINDS
w = (torch.randn(C, 512, generator=g) if cd else torch.randn(512, C, generator=g)).mul_(0.05).to(DEV)
x = torch.randn(N, 512, generator=g).to(DEV)
y = torch.randint(0, C, (N,), generator=g).to(DEV)
t = torch.zeros(1, device=DEV)
dx, dw = torch.empty_like(x), torch.empty_like(w)
def step():
    o = ops.head_forward(ctx, x, w, y, state_t=t)
    ops.head_backward(ctx, x, w, y, state_t=t, dx=dx, dw=dw)
    return o
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): o = step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
fl = 6.0 * N * 512 * C
print(f"{kind} N={N} C={C} {ms:.3f} ms per fwd+bwd, {fl / ms / 1e9:.1f} TFLOP/s algorithmic, loss {o['loss'].item():.4f}")
