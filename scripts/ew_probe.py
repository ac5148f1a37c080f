"""Diagnostic: achieved HBM bandwidth of the elementwise / reduction kernels at the ResNet-50 sizes (N=256, bf16)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
DEV = "cuda:0"; N = 256
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (H, Cw) in [(28, 256), (14, 512), (7, 1024), (4, 2048), (28, 64), (14, 128), (7, 256), (4, 512)]:
    rows = N * H * H
    y3 = torch.randn(rows, Cw, device=DEV).bfloat16(); idn = torch.randn_like(y3); out = torch.empty_like(y3)
    s3 = torch.rand(Cw, device=DEV); b3 = torch.randn(Cw, device=DEV)
    mean = torch.randn(Cw, device=DEV); inv = torch.rand(Cw, device=DEV) + 0.5
    coef = torch.randn(3, Cw, device=DEV)
    nb = ops.bn_bwd_partial_rows(rows, Cw); part = torch.empty(nb, 2, Cw, device=DEV)
    byt = rows * Cw * 2
    t1 = timeit(lambda: ops.block_merge_fwd(ops.BF16, rows, Cw, y3, s3, b3, idn, out))
    t2 = timeit(lambda: ops.bn_bwd_apply(ops.BF16, rows, Cw, y3, idn, mean, inv, coef, out, scale=s3, shift=b3, relu=True))
    t3 = timeit(lambda: ops.bn_bwd_reduce(ops.BF16, rows, Cw, y3, idn, mean, inv, part, scale=s3, shift=b3, relu=True, dz_out=out))
    prow = max(1, rows // 128); pp = torch.randn(prow, 2, Cw, device=DEV)
    g = torch.rand(Cw, device=DEV); rm = torch.zeros(Cw, device=DEV); rv = torch.ones(Cw, device=DEV)
    o = [torch.empty(Cw, device=DEV) for _ in range(4)]
    t4 = timeit(lambda: ops.bn_finalize(pp, prow, Cw, rows, g, b3, rm, rv, *o))
    t5 = timeit(lambda: ops.bn_bwd_finalize(part, nb, Cw, rows, g, mean, inv, o[0], o[1], coef))
    print(f"rows {rows:7d} C {Cw:5d}: merge {t1:6.1f}us {3*byt/t1/1e6:5.2f}TB/s | apply {t2:6.1f}us {3*byt/t2/1e6:5.2f}TB/s | "
          f"reduce {t3:6.1f}us {3*byt/t3/1e6:5.2f}TB/s | fin({prow} rows) {t4:5.1f}us | bwdfin({nb}) {t5:5.1f}us", flush=True)
