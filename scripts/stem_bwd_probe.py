"""Time the stem's backward passes at batch 256 (56x56x64): stand-alone pool gather + BN reduce + BN apply against the
fused frx_stem_bwd_reduce / frx_stem_bwd_apply.  HIP events, rotating buffer sets (operands from HBM)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
DEV = "cuda:0"; N, H, C = 256, 56, 64; Ho = 28; dt = ops.BF16; NS = 3
sets = []
for k in range(NS):
    y = torch.randn(N, H, H, C, device=DEV).bfloat16()
    dout = torch.randn(N, Ho, Ho, C, device=DEV).bfloat16()
    pooled = torch.empty(N, Ho, Ho, C, device=DEV, dtype=torch.bfloat16); arg = torch.empty(N, Ho, Ho, C, device=DEV, dtype=torch.uint8)
    sets.append(dict(y=y, dout=dout, arg=arg, pooled=pooled, dpost=torch.empty_like(y), dy=torch.empty_like(y)))
scale = torch.rand(C, device=DEV) + 0.5; shift = torch.randn(C, device=DEV) * 0.3; mean = torch.randn(C, device=DEV) * 0.2; invstd = torch.rand(C, device=DEV) + 0.7
coef = torch.randn(3 * C, device=DEV); rows = N * H * H
part = torch.zeros(max(ops.bn_bwd_partial_rows(rows, C), ops.stem_bwd_partial_rows()) * 2 * C, device=DEV)
for s in sets: ops.stem_pool_fwd(dt, N, H, H, C, s["y"], scale, shift, s["pooled"], s["arg"])
def t(fn, reps=30):
    for i in range(6): fn(sets[i % NS])
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(sets[i % NS])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
R = {
 "pool_fwd": lambda s: ops.stem_pool_fwd(dt, N, H, H, C, s["y"], scale, shift, s["pooled"], s["arg"]),
 "pool_bwd": lambda s: ops.stem_pool_bwd(dt, N, H, H, C, s["dout"], s["arg"], s["dpost"]),
 "bn_bwd_reduce(relu)": lambda s: ops.bn_bwd_reduce(dt, rows, C, s["dpost"], s["y"], mean, invstd, part, scale=scale, shift=shift, relu=True),
 "bn_bwd_apply(relu)": lambda s: ops.bn_bwd_apply(dt, rows, C, s["dpost"], s["y"], mean, invstd, coef, s["dy"], scale=scale, shift=shift, relu=True),
 "stem_bwd_reduce": lambda s: ops.stem_bwd_reduce(dt, N, H, H, C, s["dout"], s["arg"], s["y"], scale, shift, mean, invstd, part),
 "stem_bwd_apply": lambda s: ops.stem_bwd_apply(dt, N, H, H, C, s["dout"], s["arg"], s["y"], scale, shift, coef, s["dy"]),
}
for k, fn in R.items():
    print(f"{k:22s} {min(t(fn) for _ in range(3)):7.1f} us", flush=True)
