import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch, torch.nn.functional as F
from frx import ops
DEV="cuda:0"
for dtype in (0, 1):
  for (Ci, Co, k, Hi) in ((64, 64, 1, 28), (64, 256, 1, 28), (64, 64, 3, 28)):
    for N in (8, 64):
      d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, k, k, 1, k // 2)
      T = ops.TORCH_DT[dtype]
      g = torch.Generator().manual_seed(0)
      x = torch.randn(N, Hi, Hi, Ci, generator=g).to(T); w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).to(T)
      ref = F.conv2d(x.float().permute(0,3,1,2), w.float().permute(0,3,1,2), padding=k//2).permute(0,2,3,1)
      xd, wd = x.to(DEV), w.to(DEV)
      res = []
      for stats in (False, True):
        for pro in (False, True):
          y = torch.zeros(N, Hi, Hi, Co, dtype=T, device=DEV)
          part = torch.zeros(ops.conv_stat_rows(d), 2, Co, device=DEV) if stats else None
          kw = dict(in_scale=torch.ones(Ci, device=DEV), in_shift=torch.zeros(Ci, device=DEV), in_relu=False) if pro else {}
          ops.conv_fwd(d, xd, wd, y, stat_partial=part, **kw)
          torch.cuda.synchronize()
          err = (y.float().cpu() - ref).abs()
          res.append(int((~(err < 0.1 + 0.02 * ref.abs())).sum()))
      print(f"dtype {dtype} {Ci}->{Co} k{k} N={N}: bad counts [plain, pro, stats, stats+pro] = {res}")
