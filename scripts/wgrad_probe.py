"""Diagnostic: wgrad launch time per shape under different split heuristics / with the atomics replaced by stores."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
N = 256; DEV = "cuda:0"
SHAPES = [(256, 1024, 1, 1, 7), (1024, 256, 1, 1, 7), (256, 256, 3, 1, 7), (128, 512, 1, 1, 14), (512, 128, 1, 1, 14),
          (128, 128, 3, 1, 14), (512, 2048, 1, 1, 4), (512, 512, 3, 1, 4), (64, 256, 1, 1, 28), (64, 64, 3, 1, 28)]
CFGS = [("base", {}), 
        
        ("b1024 c8", {"FRX_WGRAD_BLOCKS": "1024", "FRX_WGRAD_MINCHUNKS": "8"}),
        
        ("b256 c64", {"FRX_WGRAD_BLOCKS": "256", "FRX_WGRAD_MINCHUNKS": "64"})]
def run(d, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("shape".ljust(24), *[c[0].rjust(13) for c in CFGS])
for (Ci, Co, k, st, Hi) in SHAPES:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, st, k // 2)
    x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16()
    dy = torch.randn(N, d.Ho, d.Wo, Co, device=DEV).bfloat16(); y2 = torch.randn_like(dy)
    sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
    coef = torch.randn(3, Co, device=DEV)
    dw = torch.zeros(Co, k, k, Ci, device=DEV)
    res = []
    for name, env in CFGS:
        for kk in ("FRX_WGRAD_NOATOMIC", "FRX_WGRAD_BLOCKS", "FRX_WGRAD_MINCHUNKS"): os.environ.pop(kk, None)
        os.environ.update(env)
        if k == 1:
            fn = lambda: ops.conv_wgrad_bn(d, x, dy, y2, coef, dw, in_scale=sc, in_shift=sh, in_relu=True)
        else:
            fn = lambda: ops.conv_wgrad(d, x, dy, dw, in_scale=sc, in_shift=sh, in_relu=True)
        res.append(run(d, fn))
    fl = 2.0 * N * d.Ho * d.Wo * Ci * Co * k * k
    print(f"{Ci}->{Co} k{k} H{Hi}".ljust(24), *[f"{t:7.1f}us {fl/t/1e6:4.0f}T"[:13].rjust(13) for t in res])
