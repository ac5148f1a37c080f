"""Diagnostic (needs a -DFRX_DBG_TIMES build): per-block phase timestamps of k_igemm (forward and fused dgrad)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch, numpy as np
from frx import ops, _lib
DEV = "cuda:0"; N = 256
lib = _lib.lib()
def stamps(fn_name, nblocks):
    nblocks = min(nblocks, 8192)
    buf = (C.c_longlong * (nblocks * 4))()
    rc = getattr(lib, fn_name)(buf, nblocks * 4); assert rc == 0
    return np.frombuffer(buf, dtype=np.int64).reshape(nblocks, 4).astype(np.float64) / 100.0
def report(tag, t, us):
    t = t[t[:, 3] > 0]
    t0 = t[:, 0].min()
    f, l, e = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"{tag:34s} {us:6.1f}us span {t[:,3].max()-t0:6.1f} | blocks {len(t):5d} | start med {np.median(t[:,0])-t0:5.1f} max {t[:,0].max()-t0:5.1f}"
          f" | fill {np.median(f):5.2f} loop {np.median(l):5.2f} epi {np.median(e):5.2f} (max {f.max():.1f}/{l.max():.1f}/{e.max():.1f})", flush=True)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
SH = [(64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (64, 64, 3, 1, 28), (128, 512, 1, 1, 14), (512, 128, 1, 1, 14), (128, 128, 3, 1, 14),
      (256, 1024, 1, 1, 7), (1024, 256, 1, 1, 7), (256, 256, 3, 1, 7), (512, 2048, 1, 1, 4), (2048, 512, 1, 1, 4), (512, 512, 3, 1, 4)]
for (Ci, Co, k, st, Hi) in (SH if __name__ == "__main__" else []):
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, st, k // 2)
    x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16(); w = (torch.randn(Co, k, k, Ci, device=DEV) * 0.05).bfloat16()
    y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
    sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
    part = torch.empty(ops.conv_stat_rows(d), 2, Co, device=DEV)
    fn = lambda: ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)
    us = timeit(fn)
    M = N * d.Ho * d.Wo
    bm, bn = ops._igemm_tile(d)
    nb = ((-(-M // bm) + 7) // 8 * 8) * (-(-Co // bn))
    report(f"fwd {Ci}->{Co} k{k} H{Hi} [{bm}x{bn}]", stamps("frx_debug_times_fwd", nb), us)
    if k == 1:
        # fused dgrad: dz,y [M,Co] -> dx [M,Ci], masked by the BN of the layer below (epi_y) and reduced
        wt = w.permute(3, 1, 2, 0).contiguous(); dz = torch.randn_like(y); dx = torch.empty_like(x)
        coef = torch.randn(3, Co, device=DEV); ey = torch.randn_like(x)
        esc = torch.rand(Ci, device=DEV) + 0.5; esh = torch.randn(Ci, device=DEV) * 0.1
        emu = torch.randn(Ci, device=DEV); eis = torch.rand(Ci, device=DEV) + 0.5
        ep = torch.empty(ops.conv_dgrad_stat_rows(d), 2, Ci, device=DEV)
        fn = lambda: ops.conv_dgrad_bn(d, dz, wt, dx, pro_y=y, pro_coef=coef, epi_y=ey, epi_scale=esc, epi_shift=esh,
                                       epi_mean=emu, epi_invstd=eis, epi_partial=ep)
        us = timeit(fn)
        bm, bn = ops._igemm_tile(d, True)
        nb = ((-(-M // bm) + 7) // 8 * 8) * (-(-Ci // bn))
        report(f"dgrad_bn {Ci}<-{Co} k{k} H{Hi} [{bm}x{bn}]", stamps("frx_debug_times_dgrad_bn", nb), us)
