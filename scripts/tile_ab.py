"""A/B of k_igemm tile variants per ResNet-50 conv shape (forward with BN prologue + stats epilogue, and dgrad), in one
process, interleaved rounds, HIP events.  Usage: python scripts/tile_ab.py [N]   (FRX_IGEMM_TILE is set per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DEV = "cuda:0"
VARIANTS = ["128x128x8x64", "128x64x4x64", "64x64x4x64", "64x128x4x128"]        # the instantiations FRX_IGEMM_LAUNCH knows
# (Ci, Co, k, stride, Hi)
SHAPES = [(64, 64, 3, 1, 28), (64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (128, 128, 3, 1, 14), (128, 512, 1, 1, 14), (512, 128, 1, 1, 14),
          (256, 256, 3, 1, 7), (256, 1024, 1, 1, 7), (1024, 256, 1, 1, 7), (512, 512, 3, 1, 4), (512, 2048, 1, 1, 4), (2048, 512, 1, 1, 4),
          (256, 256, 3, 2, 14), (512, 512, 3, 2, 7)]
g = torch.Generator().manual_seed(0)
def run(fn, variant):
    os.environ["FRX_IGEMM_TILE"] = variant
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
print(f"{'shape':26s} {'op':6s} " + " ".join(f"{v:>14s}" for v in VARIANTS) + "   (us, best of 7; TF/s of the best)")
for Ci, Co, k, s, Hi in SHAPES:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, s, k // 2)
    x = (torch.randn(N, Hi, Hi, Ci, generator=g)).to(DEV).bfloat16()
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).to(DEV).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
    dy = torch.randn(N, d.Ho, d.Wo, Co, generator=g).to(DEV).bfloat16()
    dx = torch.empty_like(x)
    sc, sh = torch.rand(Ci, generator=g).to(DEV) + 0.5, torch.randn(Ci, generator=g).to(DEV) * 0.1
    part = torch.zeros(2 * 4096 * max(Co, Ci) // 8 + 2 * 2048 * 2048, device=DEV)
    for op, fn in (("fwd", lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)),
                   ("dgrad", lambda: ops.conv_dgrad(d, dy, wt, dx))):
        best = {}
        ok = {}
        for v in VARIANTS:
            try:
                run(fn, v); ok[v] = True
            except Exception as e:
                ok[v] = False
        for r in range(7):
            for v in VARIANTS:
                if ok[v]:
                    t = run(fn, v); best[v] = min(best.get(v, 1e9), t)
        fl = ops.conv_flops(d)
        bv = min(best, key=best.get)
        print(f"{str((Ci, Co, k, s, Hi)):26s} {op:6s} " + " ".join(f"{best.get(v, float('nan')):14.1f}" for v in VARIANTS) + f"   best {bv} {fl / best[bv] / 1e6:.0f} TF/s", flush=True)
os.environ.pop("FRX_IGEMM_TILE", None)
