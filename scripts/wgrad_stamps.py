"""Diagnostic (needs a -DFRX_DBG_TIMES build): per-block phase timestamps of k_wgrad."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch, numpy as np
from frx import ops, _lib
DEV = "cuda:0"
lib = _lib.lib()
def stamps(nblocks):
    buf = (C.c_longlong * (nblocks * 4))()
    rc = lib.frx_debug_times(buf, nblocks * 4); assert rc == 0
    return np.frombuffer(buf, dtype=np.int64).reshape(nblocks, 4).copy()
os.environ["FRX_WGRAD_MINCHUNKS"] = "1"
for (Ci, Co, k, Hi, N, blocks, bn) in [(256, 1024, 1, 7, 256, 128, 1), (256, 1024, 1, 7, 256, 512, 1), (256, 1024, 1, 7, 32, 1024, 1),
                                       (256, 1024, 1, 7, 256, 512, 0), (256, 256, 3, 7, 256, 512, 0)]:
    os.environ["FRX_WGRAD_BLOCKS"] = str(blocks)
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, 1, k // 2)
    x = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16()
    dy = torch.randn(N, d.Ho, d.Wo, Co, device=DEV).bfloat16(); y2 = torch.randn_like(dy)
    sc = torch.rand(Ci, device=DEV) + 0.5; sh = torch.randn(Ci, device=DEV) * 0.1
    coef = torch.randn(3, Co, device=DEV)
    dw = torch.zeros(Co, k, k, Ci, device=DEV)
    if bn: fn = lambda: ops.conv_wgrad_bn(d, x, dy, y2, coef, dw, in_scale=sc, in_shift=sh, in_relu=True)
    else: fn = lambda: ops.conv_wgrad(d, x, dy, dw)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tiles = (Co // 128) * (Ci // 128) * k * k
    splits = max(1, -(-blocks // tiles))
    if splits >= 8: splits = splits // 8 * 8
    nch = -(-N * Hi * Hi // 32); cps = -(-nch // splits); splits = -(-nch // cps)
    nb = tiles * splits
    t = stamps(min(nb, 8192)).astype(np.float64) / 100.0   # us (100 MHz)
    t0 = t[:, 0].min()
    print(f"{Ci}->{Co} k{k} N{N} bn={bn}: {nb} blocks x {cps} chunks; kernel span {t[:,3].max()-t0:.1f} us")
    print("   start (min/med/max): %.1f %.1f %.1f" % (t[:,0].min()-t0, np.median(t[:,0])-t0, t[:,0].max()-t0))
    for a, b, nm in ((0, 1, "fill"), (1, 2, "loop"), (2, 3, "epilogue")):
        dd = t[:, b] - t[:, a]
        print(f"   {nm:9s} min/med/max: {dd.min():.2f} {np.median(dd):.2f} {dd.max():.2f} us")
