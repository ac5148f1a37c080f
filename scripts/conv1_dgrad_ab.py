"""Time the conv1-type input gradients (1x1, Co -> Ci = 4 Co, K = Co: BN-backward prologue, skip-gradient addend, merge-ReLU
mask bits, BN-backward statistics epilogue) of the four ResNet-50 stages at batch 256, rotating over NSETS tensor sets so
the operands come from HBM as they do inside a training step.  A/B builds: FRX_LIB=path/to/libfrx.so."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops, _lib
if os.environ.get("FRX_LIB"):
    _lib.load_library(os.path.join(ROOT, os.environ["FRX_LIB"]))
DEV = "cuda:0"; N = 256; NSETS = int(os.environ.get("NSETS", 3))
SH = [(256, 64, 28), (512, 128, 14), (1024, 256, 7), (2048, 512, 4)]
def make(Ci, Co, Hi):
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    M = N * Hi * Hi
    sets = []
    for _ in range(NSETS):
        s = dict(dz=torch.randn(M, Co, device=DEV).bfloat16(), y=torch.randn(M, Co, device=DEV).bfloat16(),
                 add=torch.randn(M, Ci, device=DEV).bfloat16(), ey=torch.randn(M, Ci, device=DEV).bfloat16(),
                 bits=torch.randint(0, 256, (M * Ci // 8,), device=DEV, dtype=torch.uint8), dx=torch.empty(M, Ci, device=DEV, dtype=torch.bfloat16))
        sets.append(s)
    w = (torch.randn(Ci, 1, 1, Co, device=DEV) * 0.05).bfloat16()          # [C][R][S][K] = transposed filter
    coef = torch.randn(3, Co, device=DEV); emu = torch.randn(Ci, device=DEV); eis = torch.rand(Ci, device=DEV) + 0.5
    ep = torch.empty(ops.conv_dgrad_stat_rows(d), 2, Ci, device=DEV)
    def run(i):
        s = sets[i % NSETS]
        ops.conv_dgrad_bn(d, s["dz"], w, s["dx"], addend=s["add"], pro_y=s["y"], pro_coef=coef, epi_y=s["ey"],
                          epi_out_bits=s["bits"], epi_mean=emu, epi_invstd=eis, epi_partial=ep)
    nbytes = M * Co * 2 * 2 + M * Ci * 2 * 3 + M * Ci // 8
    return run, nbytes, 2.0 * M * Ci * Co
def timeit(run, reps=30):
    for i in range(6): run(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def stamps(nblocks):
    import ctypes as C, numpy as np
    fn = getattr(_lib.lib(), "frx_debug_times_dgrad_bn", None)       # only in a -DFRX_DBG_TIMES build
    if fn is None:
        return
    nblocks = min(nblocks, 8192)
    buf = (C.c_longlong * (nblocks * 4))()
    assert fn(buf, nblocks * 4) == 0
    t = np.frombuffer(buf, dtype=np.int64).reshape(nblocks, 4).astype(np.float64) / 100.0
    t = t[t[:, 3] > 0]
    t = t[t[:, 0] > t[:, 0].max() - 1000.0]          # the last launch only (earlier, larger grids leave their stamps behind)
    t0 = t[:, 0].min()
    f, l, e = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"    stamps: span {t[:,3].max()-t0:6.1f} us | blocks {len(t):5d} | start med {np.median(t[:,0])-t0:5.1f} max {t[:,0].max()-t0:5.1f}"
          f" | fill {np.median(f):5.2f} loop {np.median(l):5.2f} epi {np.median(e):5.2f} (max {f.max():.1f}/{l.max():.1f}/{e.max():.1f})", flush=True)
    # how many blocks are in flight over time (quartiles of the span)
    for q in (0.1, 0.3, 0.5, 0.7, 0.9):
        tq = t0 + q * (t[:, 3].max() - t0)
        print(f"      t={q:.1f}: {int(((t[:,0] <= tq) & (t[:,3] > tq)).sum())} blocks resident", flush=True)
tot = 0.0
for (Ci, Co, Hi) in SH:
    run, nb, fl = make(Ci, Co, Hi)
    us = min(timeit(run) for _ in range(3))
    run(0); torch.cuda.synchronize()
    stamps(8192)
    cnt = {28: 2, 14: 3, 7: 5, 4: 2}[Hi]
    tot += us * cnt
    print(f"conv1 dgrad {Ci}<-{Co} H{Hi}: {us:7.1f} us  {nb / us * 1e-3:6.0f} GB/s  {fl / us * 1e-6:6.1f} TF/s   (x{cnt} per step)", flush=True)
print(f"sum over the 12 identity blocks: {tot:.0f} us")
