"""A/B of the LDS-DMA staging of k_igemm (FRX_IGEMM_DMA = 0 / 3 / 4, read per launch) on the prologue-free launches of a
ResNet-50 step at batch N: forward + statistics of the conv1 / projection type, plain 3x3 forward (as it would run on a
pre-normalised input), and the 3x3 / 1x1 input gradients.  One process, interleaved rounds, HIP events; results of the
variants are compared bit for bit (the staging path must not change a single output).
Usage: python scripts/dma_ab.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DEV = "cuda:0"
VARIANTS = ["0", "4", "8"]      # (3 / 6 on the 64x128 tile)
# (Ci, Co, k, stride, Hi)
SHAPES = [(256, 64, 1, 1, 28), (256, 128, 1, 1, 28), (512, 128, 1, 1, 14), (128, 128, 3, 1, 14), (1024, 256, 1, 1, 7), (256, 256, 3, 1, 7),
          (2048, 512, 1, 1, 4), (512, 512, 3, 1, 4), (256, 512, 1, 2, 28), (512, 1024, 1, 2, 14), (64, 64, 3, 1, 28), (128, 128, 3, 2, 28),
          (256, 256, 3, 2, 14)]
g = torch.Generator().manual_seed(0)


def run(fn, variant):
    os.environ["FRX_IGEMM_DMA"] = variant
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


print(f"{'shape':26s} {'op':10s} {'tile':>9s} " + " ".join(f"{'dma=' + v:>9s}" for v in VARIANTS) + "   (us, best of 9)   TF/s of the best")
for Ci, Co, k, s, Hi in SHAPES:
    d = ops.conv_desc(ops.BF16, N, Hi, Hi, Ci, Co, k, k, s, k // 2)
    x = (torch.randn(N, Hi, Hi, Ci, generator=g)).to(DEV).bfloat16()
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).to(DEV).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    y = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=torch.bfloat16)
    dy = torch.randn(N, d.Ho, d.Wo, Co, generator=g).to(DEV).bfloat16()
    dx = torch.empty_like(x)
    ey = torch.randn(N, Hi, Hi, Ci, generator=g).to(DEV).bfloat16()
    esc, esh = torch.rand(Ci, generator=g).to(DEV) + 0.5, torch.randn(Ci, generator=g).to(DEV) * 0.1
    emu, eis = torch.randn(Ci, generator=g).to(DEV) * 0.1, torch.rand(Ci, generator=g).to(DEV) + 0.5
    part = torch.zeros(2 * 4096 * max(Co, Ci) // 8 + 2 * 2048 * 2048, device=DEV)
    ops_ = [("fwd+stats", False, lambda: ops.conv_fwd(d, x, w, y, stat_partial=part), lambda: (y, part)),
            ("dgrad+bn", True, lambda: ops.conv_dgrad_bn(d, dy, wt, dx, epi_y=ey, epi_scale=esc, epi_shift=esh, epi_mean=emu, epi_invstd=eis,
                                                        epi_partial=part), lambda: (dx, part))]
    for op, dg, fn, outs in ops_:
        best, res = {}, {}
        for v in VARIANTS:
            part.zero_()
            run(fn, v)
            res[v] = [t.clone() for t in outs()]
        for v in VARIANTS[1:]:
            same = all(torch.equal(a, b) for a, b in zip(res[v], res["0"]))
            if not same:
                err = max((a.float() - b.float()).abs().max().item() for a, b in zip(res[v], res["0"]))
                print(f"  !! dma={v} differs from the register-staged result: max |diff| {err:.3e}")
        for r in range(9):
            for v in VARIANTS:
                t = run(fn, v); best[v] = min(best.get(v, 1e9), t)
        fl = ops.conv_flops(d)
        bv = min(best, key=best.get)
        print(f"{str((Ci, Co, k, s, Hi)):26s} {op:10s} {str(ops._igemm_tile(d, dg)):>9s} " + " ".join(f"{best[v]:9.1f}" for v in VARIANTS) +
              f"   best dma={bv} {fl / best[bv] / 1e6:.0f} TF/s", flush=True)
os.environ.pop("FRX_IGEMM_DMA", None)
