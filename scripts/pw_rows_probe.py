"""Diagnostic: the conv1-type input gradients of ResNet-50 at batch 256 (bf16, masked-statistics epilogue, addend) on
csrc/pw_rows.hip against k_igemm (FRX_PW_ROWS=0), stand-alone launches; with a -DFRX_DBG_TIMES build (FRX_LIB=...)
also the per-block phase stamps of the row-resident kernel: set-up / items / flush."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch, numpy as np
from frx import ops, _lib
DEV = "cuda:0"; N = 256; R = 8
lib = _lib.lib()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
flush = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
for (Ci, Co, Hi) in [(256, 64, 28), (512, 128, 14), (1024, 256, 7)]:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    M = N * Hi * Hi
    dz = torch.randn(N, Hi, Hi, Co, device=DEV).bfloat16(); y = torch.randn_like(dz)
    wt = (torch.randn(Ci, 1, 1, Co, device=DEV) * Co ** -0.5).bfloat16()
    coef = torch.randn(3 * Co, device=DEV)
    ey = torch.randn(N, Hi, Hi, Ci, device=DEV).bfloat16(); add = torch.randn_like(ey)
    bits = torch.randint(0, 256, (M * Ci // 8,), device=DEV, dtype=torch.uint8)
    emu = torch.randn(Ci, device=DEV); eis = torch.rand(Ci, device=DEV) + 0.5
    tout = torch.zeros(R, 2, Ci, device=DEV); dx = torch.empty_like(ey)
    fn = lambda: ops.conv_dgrad_bn(d, dz, wt, dx, addend=add, pro_y=y, pro_coef=coef, epi_y=ey, epi_out_bits=bits, epi_mean=emu,
                                   epi_invstd=eis, epi_totals=tout, epi_replicas=R)
    res = []
    cfgs = [None] + (sys.argv[1:] or [""])
    for cfg in cfgs:
        os.environ["FRX_PW_ROWS"] = "0" if cfg is None else "3"
        if cfg: os.environ["FRX_PWR_RING"] = cfg
        else: os.environ.pop("FRX_PWR_RING", None)
        try:
            fn()
        except Exception as e:
            res.append((float("nan"), float("nan"))); continue
        warm = timeit(fn)
        cold = []
        for _ in range(5):
            flush.fill_(1); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        res.append((warm, float(np.median(cold))))
        if os.environ.get("PROBE_WARM_LAST"): timeit(fn, 3)
    line = f"dgrad {Ci}<-{Co} H{Hi}: k_igemm {res[0][0]:6.1f} us warm {res[0][1]:6.1f} cold | rows " + " | ".join(
        f"[{c or 'default'}] {r[0]:6.1f} warm {r[1]:6.1f} cold" for c, r in zip(cfgs[1:], res[1:]))
    if hasattr(lib, "frx_debug_times_pw_rows"):
        items = -(-M // 64) * (Ci // 128)
        nb = min(items, 256)
        buf = (C.c_longlong * (nb * 4))()
        assert lib.frx_debug_times_pw_rows(buf, nb * 4) == 0
        t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 4).astype(np.float64) / 100.0
        per = t / (items / nb)
        line += f" | per item (compute wave 0, us): barrier wait {np.median(per[:,0]):5.2f} row-block switch {np.median(per[:,1]):5.2f} mfma {np.median(per[:,2]):5.2f} epilogue {np.median(per[:,3]):5.2f}"
    print(line, flush=True)
    # the conv3-type forward of the same block: Co -> Ci here (middle width -> block width)
    df = ops.conv_desc(ops.BF16, N, Hi, Hi, Co, Ci, 1, 1, 1, 0)
    xin = torch.randn(N, Hi, Hi, Co, device=DEV).bfloat16(); wf = (torch.randn(Ci, 1, 1, Co, device=DEV) * Co ** -0.5).bfloat16()
    sc = torch.rand(Co, device=DEV) + 0.5; sh = torch.randn(Co, device=DEV) * 0.1
    tin = torch.zeros(R, 2, Co, device=DEV); tin[0, 0] = 1.0; tin[0, 1] = float(M)
    gam = torch.rand(Co, device=DEV) + 0.5; bet = torch.randn(Co, device=DEV) * 0.1
    yout = torch.empty(N, Hi, Hi, Ci, device=DEV, dtype=torch.bfloat16); tf = torch.zeros(R, 2, Ci, device=DEV)
    bnin = ops.bn_tot(tin, R, M, gam, beta=bet)
    ffn = lambda: ops.conv_fwd_tot(df, xin, wf, yout, in_bn=bnin, in_relu=True, stat_totals=tf, stat_replicas=R)
    r = []
    for mode in ("0", "3"):
        os.environ["FRX_PW_ROWS"] = mode
        warm = timeit(ffn); cold = []
        for _ in range(5):
            flush.fill_(1); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ffn(); e1.record(); torch.cuda.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        r.append((warm, float(np.median(cold))))
    print(f"fwd   {Co}->{Ci} H{Hi}: k_igemm {r[0][0]:6.1f} us warm {r[0][1]:6.1f} cold | rows {r[1][0]:6.1f} warm {r[1][1]:6.1f} cold", flush=True)
