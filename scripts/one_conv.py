"""ONE convolution shape through the product path, repeated, for rocprofv3 --pmc / --kernel-trace runs on a single kernel.
Usage: python scripts/one_conv.py Ci Co k stride Hi [kind] [reps] [N]
kind: fwd (BN+ReLU prologue + statistics: the production forward of a 3x3 / conv3), fwd0 (prologue-free + statistics: conv1 type),
      dgrad (BN-backward prologue + statistics epilogue), dgrad0 (prologue-free input gradient)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
Ci, Co, k, s, Hi = (int(v) for v in sys.argv[1:6])
kind = sys.argv[6] if len(sys.argv) > 6 else "fwd"
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
N = int(sys.argv[8]) if len(sys.argv) > 8 else 256
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, s, k // 2)
x = torch.randn(N, Hi, Hi, Ci, generator=g).to(DEV).bfloat16()
w = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).to(DEV).bfloat16()
wt = w.permute(3, 1, 2, 0).contiguous()
y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
dy = torch.randn(N, d.Ho, d.Wo, Co, generator=g).to(DEV).bfloat16()
yy = torch.randn(N, d.Ho, d.Wo, Co, generator=g).to(DEV).bfloat16()
dx = torch.empty_like(x)
ey = torch.randn(N, Hi, Hi, Ci, generator=g).to(DEV).bfloat16()
sc, sh = torch.rand(Ci, generator=g).to(DEV) + 0.5, torch.randn(Ci, generator=g).to(DEV) * 0.1
esc, esh = torch.rand(Ci, generator=g).to(DEV) + 0.5, torch.randn(Ci, generator=g).to(DEV) * 0.1
emu, eis = torch.randn(Ci, generator=g).to(DEV) * 0.1, torch.rand(Ci, generator=g).to(DEV) + 0.5
coef = torch.randn(3, Co, generator=g).to(DEV)
part = torch.zeros(2 * 4096 * max(Co, Ci) // 8 + 2 * 2048 * 2048, device=DEV)
fns = {
    "fwd": lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part),
    "fwd0": lambda: ops.conv_fwd(d, x, w, y, stat_partial=part),
    "dgrad": lambda: ops.conv_dgrad_bn(d, dy, wt, dx, pro_y=yy, pro_coef=coef, epi_y=ey, epi_scale=esc, epi_shift=esh, epi_mean=emu,
                                       epi_invstd=eis, epi_partial=part),
    "dgrad0": lambda: ops.conv_dgrad_bn(d, dy, wt, dx, epi_y=ey, epi_scale=esc, epi_shift=esh, epi_mean=emu, epi_invstd=eis, epi_partial=part),
}
fn = fns[kind]
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(reps):
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
fl = ops.conv_flops(d)
print(f"{kind} {(Ci, Co, k, s, Hi)} tile {ops._igemm_tile(d, kind.startswith('dgrad'))}: best {min(ts):.1f} us, median {sorted(ts)[len(ts) // 2]:.1f} us, {fl / min(ts) / 1e6:.0f} TF/s")
