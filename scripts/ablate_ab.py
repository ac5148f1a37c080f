"""Ablation of k_igemm<bf16,128,128> (8 waves) on a few conv shapes: which part of the K loop costs the time?
Variants via FRX_IGEMM_TILE=128x128x8x64x<code>: 3 = as shipped; 101 no global loads in the loop; 102 no register->LDS
commit (no prologue, no ds_write); 103 both; 104 no MFMA; 108 no barrier; 111 = 101+102+108; 114 = 102+104+108.
(Results of ablated variants are garbage by construction: timing only.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
N, DEV = 256, "cuda:0"
CODES = [3, 101, 102, 103, 104, 108, 111, 114]
SHAPES = [(256, 256, 3, 1, 7), (128, 128, 3, 1, 14), (64, 64, 3, 1, 28), (1024, 256, 1, 1, 7), (256, 1024, 1, 1, 7), (64, 256, 1, 1, 28), (512, 512, 3, 1, 4)]
g = torch.Generator().manual_seed(0)
def run(fn, code):
    os.environ["FRX_IGEMM_TILE"] = f"128x128x8x64x{code}"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
print(f"{'shape':26s} {'op':6s} " + " ".join(f"{c:>8d}" for c in CODES))
for Ci, Co, k, s, Hi in SHAPES:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, s, k // 2)
    x = torch.randn(N, Hi, Hi, Ci, generator=g).to(DEV).bfloat16()
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).to(DEV).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
    dy = torch.randn(N, d.Ho, d.Wo, Co, generator=g).to(DEV).bfloat16()
    dx = torch.empty_like(x)
    sc, sh = torch.rand(Ci, generator=g).to(DEV) + 0.5, torch.randn(Ci, generator=g).to(DEV) * 0.1
    part = torch.zeros(2 * 2048 * 2048, device=DEV)
    for op, fn in (("fwd", lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)),
                   ("dgrad", lambda: ops.conv_dgrad(d, dy, wt, dx))):
        best = {}
        for r in range(6):
            for c in CODES:
                t = run(fn, c); best[c] = min(best.get(c, 1e9), t) if r else 1e9
        print(f"{str((Ci, Co, k, s, Hi)):26s} {op:6s} " + " ".join(f"{best[c]:8.1f}" for c in CODES), flush=True)
os.environ.pop("FRX_IGEMM_TILE", None)
