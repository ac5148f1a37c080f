import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch, torch.nn.functional as F
from frx import ops
DEV="cuda:0"
for dtype in (0, 1):
  for N in (3, 8, 16, 32):
    g = torch.Generator().manual_seed(0)
    img = torch.rand(N, 3, 112, 112, generator=g) * 2 - 1
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    hp, wp = ops.stem_padded_dims(112, 112)
    xin = torch.empty(N, hp, wp, 4, dtype=ops.TORCH_DT[dtype], device=DEV)
    ops.input_prep(dtype, img.to(DEV), xin)
    wk = torch.zeros(64, 7, 8, 4); wk[:, :, :7, :3] = w.permute(0, 2, 3, 1); wk = wk.to(ops.TORCH_DT[dtype])
    d = ops.conv_desc(dtype, N, 112, 112, 3, 64, 7, 7, 2, 3, stem=True)
    y = torch.zeros(N, 56, 56, 64, dtype=xin.dtype, device=DEV)
    part = torch.zeros(ops.conv_stat_rows(d), 2, 64, device=DEV)
    ops.conv_fwd(d, xin, wk.to(DEV), y, stat_partial=part)
    ref = F.conv2d(img, w, stride=2, padding=3).permute(0, 2, 3, 1)
    print(dtype, N, "err", (y.float().cpu() - ref).abs().max().item(), "ynorm", y.float().norm().item(), "refnorm", ref.norm().item())
print("---- detail N=16 bf16")
dtype, N = 1, 16
g = torch.Generator().manual_seed(0)
img = torch.rand(N, 3, 112, 112, generator=g) * 2 - 1
w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
xin = torch.empty(N, hp, wp, 4, dtype=torch.bfloat16, device=DEV)
ops.input_prep(dtype, img.to(DEV), xin)
print("xin finite", torch.isfinite(xin.float()).all().item(), "xin absmax", xin.float().abs().max().item())
wk = torch.zeros(64, 7, 8, 4); wk[:, :, :7, :3] = w.permute(0, 2, 3, 1); wk = wk.bfloat16().to(DEV)
d = ops.conv_desc(dtype, N, 112, 112, 3, 64, 7, 7, 2, 3, stem=True)
for trial in range(3):
    y = torch.zeros(N, 56, 56, 64, dtype=torch.bfloat16, device=DEV)
    ops.conv_fwd(d, xin, wk, y)
    bad = ~torch.isfinite(y.float())
    ref = F.conv2d(img, w, stride=2, padding=3).permute(0, 2, 3, 1)
    err = (y.float().cpu() - ref).abs()
    big = (err > 0.05)
    print("trial", trial, "nonfinite", bad.sum().item(), "big errs", big.sum().item(), "first bad idx", big.nonzero()[:3].tolist())
print("---- stats variant")
for N in (8, 16, 32):
    g = torch.Generator().manual_seed(0)
    img = torch.rand(N, 3, 112, 112, generator=g) * 2 - 1
    xin = torch.empty(N, hp, wp, 4, dtype=torch.bfloat16, device=DEV)
    ops.input_prep(1, img.to(DEV), xin)
    d = ops.conv_desc(1, N, 112, 112, 3, 64, 7, 7, 2, 3, stem=True)
    ref = F.conv2d(img, w, stride=2, padding=3).permute(0, 2, 3, 1)
    for trial in range(2):
        y = torch.zeros(N, 56, 56, 64, dtype=torch.bfloat16, device=DEV)
        part = torch.zeros(ops.conv_stat_rows(d), 2, 64, device=DEV)
        ops.conv_fwd(d, xin, wk, y, stat_partial=part)
        torch.cuda.synchronize()
        err = (y.float().cpu() - ref).abs()
        big = ~(err < 0.05)
        print("N", N, "trial", trial, "bad", big.sum().item(), "first", big.nonzero()[:2].tolist(), "rows", ops.conv_stat_rows(d))
