"""GPU parity: HIP head (cosine GEMM + margin + CE + top-k, forward and backward) and the
pair-cosine kernel, through the C ABI, against the oracle and the reference's golden vectors."""
import os

import zlib

import numpy as np
import pytest
import torch

from oracle import heads as H
from oracle import verify as V

pytestmark = pytest.mark.gpu
from test_oracle_heads import KINDS as GOLDEN_KINDS, hyper_for

KINDS = {"arcface": H.ARC, "cosface": H.COS, "sphereface": H.SPHERE, "curricular": H.CURR}
LOGIT_TOL = 1e-3          # north-star tolerance on logits (cosine x 64)


def _run(kind, x, w, y, hyper, t0=0.0, lamb=0.0, want_logits=True):
    from frx import ops
    dev = torch.device("cuda:0")
    N, D = x.shape
    Cc = w.shape[0] if H.weight_is_cd(kind) else w.shape[1]
    ctx = ops.HeadContext(kind, N, D, Cc, hyper.s, float(hyper.m), hyper.momentum, device=dev,
                          flags=(1 if (kind == H.ARC and hyper.easy_margin) else 0))
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    wd = torch.from_numpy(np.ascontiguousarray(w)).to(dev)
    yd = torch.from_numpy(np.asarray(y).astype(np.int64)).to(dev)
    t = torch.full((1,), float(t0), device=dev)
    o = ops.head_forward(ctx, xd, wd, yd, state_t=t, lamb=lamb, want_logits=want_logits)
    dx, dw = ops.head_backward(ctx, xd, wd, yd, state_t=t)
    torch.cuda.synchronize()
    return o, dx.cpu().numpy(), dw.cpu().numpy(), float(t.item())


@pytest.mark.parametrize("name,N,Cc", [("arcface", 32, 100), ("cosface", 37, 1000), ("mv_am", 64, 333), ("mv_arc", 16, 64),
                                       ("arcface", 256, 10575), ("cosface", 256, 10575), ("arcface", 64, 85742)])
def test_fused_forward_equals_the_row_sweep(name, N, Cc):
    """Round 4 (north_star: cosine GEMM + fused row-max / exp / sum epilogue): without [N, C] outputs the forward of ARC / COS /
    MV takes the fused path -- target cosines as N dot products, the GEMM's epilogue leaves per-tile (max, sum-exp, rank)
    partials, one combine launch closes loss / lse / top-k -- against the row sweep over the stored cosines that the same
    call runs when cos_s / logits are asked for: lse and loss to fp32 rounding of the sums, top-k counts identical, and the
    backward (which reads the cosines the GEMM wrote, the target element being the dot product's own value) identical."""
    from frx import ops
    kinds = {"arcface": H.ARC, "cosface": H.COS, "mv_am": H.MV_AM, "mv_arc": H.MV_ARC}
    kind = kinds[name]
    hy = hyper_for(name) if name in ("arcface", "cosface") else None
    rng = np.random.RandomState(N + Cc)
    x = rng.randn(N, 512).astype(np.float32)
    w = (rng.randn(Cc, 512) if H.weight_is_cd(kind) else rng.randn(512, Cc)).astype(np.float32) * 0.05
    y = rng.randint(0, Cc, N)
    y[0], y[-1] = 0, Cc - 1
    if H.weight_is_cd(kind):                                     # a few rows with a REAL target (cosine near 0.7: the margin branch)
        for i in range(0, N, 5):
            w[y[i]] = 0.7 * x[i] / np.linalg.norm(x[i]) * np.linalg.norm(w[y[i]]) + 0.3 * w[y[i]]
    dev = torch.device("cuda:0")
    s_, m_ = (hy.s, float(hy.m)) if hy is not None else (32.0, 0.35)
    p = (1.12,) if hy is None else ()
    res = []
    for want in (True, False):
        ctx = ops.HeadContext(kind, N, 512, Cc, s_, m_, 0.01, device=dev, p=p)
        xd, wd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(y.astype(np.int64)).to(dev)
        o = ops.head_forward(ctx, xd, wd, yd, want_logits=want)
        dx, dw = ops.head_backward(ctx, xd, wd, yd)
        torch.cuda.synchronize()
        res.append((o, dx.clone(), dw.clone()))
    (oa, dxa, dwa), (ob, dxb, dwb) = res
    assert ob["logits"] is None and oa["logits"] is not None
    assert abs(oa["loss"].item() - ob["loss"].item()) < 2e-5 * abs(oa["loss"].item())
    assert (oa["lse"] - ob["lse"]).abs().max().item() < 2e-5 * oa["lse"].abs().max().item()
    assert torch.equal(oa["topk"], ob["topk"]), (oa["topk"], ob["topk"])
    assert torch.equal(oa["norms"], ob["norms"])
    # an independent top-k from the full outputs of the row-sweep run
    top5 = oa["cos_s"].topk(5, dim=1).indices
    yd = torch.from_numpy(y.astype(np.int64)).to(dev)
    assert int(ob["topk"][0]) == int((top5[:, 0] == yd).sum()) and int(ob["topk"][1]) == int((top5 == yd[:, None]).any(1).sum())
    for a, b in ((dxa, dxb), (dwa, dwb)):
        assert ((a - b).norm() / (a.norm() + 1e-30)).item() < 1e-5


def _ill(kind, cos_s, y, norms):
    ty = cos_s[np.arange(len(y)), y] / (64.0 if kind != H.SPHERE else norms.reshape(-1))
    return (np.abs(ty) > 1 - 1e-5) & (kind in (H.ARC, H.CURR))


@pytest.mark.parametrize("name", list(GOLDEN_KINDS))
@pytest.mark.parametrize("tag", ["fresh", "warm"])
def test_head_vs_reference_golden(golden_dir, name, tag):
    g = np.load(os.path.join(golden_dir, f"heads_{name}.npz"))
    kind = GOLDEN_KINDS[name]
    hy = hyper_for(name)
    lamb = float(g[f"{tag}_lamb"])
    o, dx, dw, t_after = _run(kind, g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_y"], hy,
                              t0=float(g[f"{tag}_pre_t"]), lamb=lamb)
    y = g[f"{tag}_y"]
    ill = _ill(kind, g[f"{tag}_cos_s"], y, g[f"{tag}_norms"])
    ok = ~ill
    np.testing.assert_allclose(o["cos_s"].cpu().numpy(), g[f"{tag}_cos_s"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(o["logits"].cpu().numpy()[ok], g[f"{tag}_logits"][ok], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(o["norms"].cpu().numpy(), g[f"{tag}_norms"].reshape(-1), rtol=1e-5)
    assert abs(o["loss"].item() - float(g[f"{tag}_loss"])) < (1e-3 if ok.all() else 1e-2)
    n = len(y)
    top = o["topk"].cpu().numpy()
    assert 100.0 * top[0] / n == pytest.approx(float(g[f"{tag}_acc1"]), abs=1e-4)
    assert 100.0 * top[1] / n == pytest.approx(float(g[f"{tag}_acc5"]), abs=1e-4)
    sx, sw = np.nanmax(np.abs(g[f"{tag}_dx"])), np.nanmax(np.abs(g[f"{tag}_dw"]))
    np.testing.assert_allclose(dx[ok], g[f"{tag}_dx"][ok], atol=1e-3 * sx, rtol=0)
    wc = (lambda a: a) if H.weight_is_cd(kind) else (lambda a: a.T)
    okc = np.ones(wc(dw).shape[0], dtype=bool)
    okc[y[ill]] = False
    np.testing.assert_allclose(wc(dw)[okc], wc(g[f"{tag}_dw"])[okc], atol=1e-3 * sw, rtol=0)
    if kind == H.CURR:
        assert t_after == pytest.approx(float(g[f"{tag}_post_t"]), abs=1e-6)


@pytest.mark.parametrize("name", list(KINDS))
@pytest.mark.parametrize("shape", [(32, 512, 100), (48, 512, 1000), (7, 64, 37)])
def test_head_vs_oracle_seeded(name, shape):
    """Config-1 head shape (N=32, C=100) plus ragged shapes (odd C, N not a tile multiple)."""
    kind = KINDS[name]
    N, D, Cc = shape
    rng = np.random.RandomState(zlib.crc32(repr((name, shape)).encode()) % 2**31)      # (hash() of a str is salted per process)
    wshape = (Cc, D) if H.weight_is_cd(kind) else (D, Cc)
    w = (rng.randn(*wshape) * 0.05).astype(np.float32)
    y = rng.randint(0, Cc, N)
    x = rng.randn(N, D).astype(np.float32)
    wc = w if H.weight_is_cd(kind) else w.T
    for i in range(0, N, 3):            # every third row sits near its class centre
        x[i] = wc[y[i]] / np.linalg.norm(wc[y[i]]) * 4 + 0.3 * rng.randn(D)
    hy = H.HeadHyper.default(kind)
    st = H.HeadState(iter=6, t=0.123)
    ref = H.head_forward_backward(kind, x, w, y, hy, st, dtype=np.float64)
    o, dx, dw, t_after = _run(kind, x, w, y, hy, t0=0.123, lamb=st.lamb)
    ok = np.ones(ref.logits.shape, dtype=bool)
    if kind == H.CURR:
        # CurricularFace's hard-example mask cos > cos(theta_y + m) (criterion.py:560-565) is a step: of 11 M cosines a
        # handful sit within fp32 rounding of their row's threshold and take the other branch than float64 does (the
        # reference's own fp32 run has the same freedom).  They are left out of the logit comparison, and counted.
        ty = ref.cos_s[np.arange(N), y] / hy.s
        ctm = ty * np.cos(hy.m) - np.sqrt(np.maximum(0.0, 1 - ty * ty)) * np.sin(hy.m)
        ok = np.abs(ref.cos_s / hy.s - ctm[:, None]) > 3e-6
        assert (~ok).sum() <= max(2, 1e-5 * ok.size)
    np.testing.assert_allclose(o["logits"].cpu().numpy()[ok], ref.logits[ok], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(o["cos_s"].cpu().numpy(), ref.cos_s, atol=LOGIT_TOL, rtol=0)
    assert abs(o["loss"].item() - ref.loss) < 1e-3
    np.testing.assert_allclose(o["lse"].cpu().numpy(), ref.lse, atol=1e-3)
    assert tuple(o["topk"].cpu().numpy()) == (ref.top1, ref.top5)
    np.testing.assert_allclose(dx, ref.dx, atol=1e-3 * np.abs(ref.dx).max(), rtol=0)
    np.testing.assert_allclose(dw, ref.dw, atol=1e-3 * np.abs(ref.dw).max(), rtol=0)
    if kind == H.CURR:
        assert t_after == pytest.approx(st.t, abs=1e-6)


@pytest.mark.parametrize("name,N,Cc", [("arcface", 256, 10575), ("cosface", 256, 10575), ("curricular", 128, 85000)])
def test_head_full_size_properties(name, N, Cc):
    """BASELINE sizes: size-independent properties instead of an O(N*C*D) oracle run:
    softmax rows sum to 1 (via lse), loss == mean(lse - z_y), rows of dC sum to ~0 for the
    unmargined part, and a sampled set of logits equals a float64 dot product."""
    from frx import ops
    kind = KINDS[name]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    D = 512
    wshape = (Cc, D) if H.weight_is_cd(kind) else (D, Cc)
    w = torch.randn(*wshape, generator=g) * 0.05
    x = torch.randn(N, D, generator=g)
    y = torch.randint(0, Cc, (N,), generator=g)
    hy = H.HeadHyper.default(kind)
    ctx = ops.HeadContext(kind, N, D, Cc, hy.s, float(hy.m), hy.momentum, device=dev)
    t = torch.zeros(1, device=dev)
    o = ops.head_forward(ctx, x.to(dev), w.to(dev), y.to(dev), state_t=t, want_logits=True)
    z = o["logits"].double().cpu()
    lse = torch.logsumexp(z, dim=1)
    np.testing.assert_allclose(o["lse"].cpu().numpy(), lse.numpy(), atol=1e-3)
    loss = (lse - z[torch.arange(N), y]).mean().item()
    assert abs(o["loss"].item() - loss) < 1e-3
    # sampled cosines vs float64
    wc = (w if H.weight_is_cd(kind) else w.t()).double()
    xn = torch.nn.functional.normalize(x.double(), dim=1)
    idx = torch.randint(0, Cc, (64,), generator=g)
    cos = xn @ torch.nn.functional.normalize(wc[idx], dim=1).t()
    got = o["cos_s"].cpu().double()[:, idx] / hy.s
    np.testing.assert_allclose(got.numpy(), cos.clamp(-1, 1).numpy(), atol=1e-3 / 64)
    # top-k against torch.topk on the materialised scaled cosines
    cs = o["cos_s"].cpu()
    _, pred = cs.topk(5, 1, True, True)
    hit = pred.eq(y.view(-1, 1))
    assert int(hit[:, :1].sum()) == int(o["topk"][0]) and int(hit.sum()) == int(o["topk"][1])
    # gradients: finite, and dx orthogonal to x for the scale-invariant heads (d/ds L(s*x) = 0)
    dx, dw = ops.head_backward(ctx, x.to(dev), w.to(dev), y.to(dev), state_t=t)
    assert torch.isfinite(dx).all() and torch.isfinite(dw).all()
    rad = (dx.cpu() * x).sum(1).abs().max().item()
    assert rad < 1e-4 * dx.abs().max().item() * x.norm(dim=1).max().item() + 1e-6


@pytest.mark.parametrize("name,N,Cc", [("curricular", 128, 85742), ("arcface", 64, 85742)])
def test_head_ragged_85742_vs_float64_oracle(name, N, Cc):
    """configs[3]'s second width (85 742 identities = 2 x 43 x 997, no tile multiple) at full size,
    both weight layouts, against the float64 oracle: logits / cosines within the north-star 1e-3, loss, lse, top-k,
    dX and dW within 1e-3 of their scale, CurricularFace's EMA."""
    kind = KINDS[name]
    D = 512
    rng = np.random.RandomState(85742 + N)
    wshape = (Cc, D) if H.weight_is_cd(kind) else (D, Cc)
    w = (rng.randn(*wshape) * 0.05).astype(np.float32)
    y = rng.randint(0, Cc, N)
    y[:4] = [0, Cc - 1, Cc - 2, 85696]          # first / last columns, and the first column of the ragged last tile
    x = rng.randn(N, D).astype(np.float32)
    wc = w if H.weight_is_cd(kind) else w.T
    for i in range(0, N, 3):
        x[i] = wc[y[i]] / np.linalg.norm(wc[y[i]]) * 4 + 0.3 * rng.randn(D)
    hy = H.HeadHyper.default(kind)
    st = H.HeadState(iter=6, t=0.123)
    ref = H.head_forward_backward(kind, x, w, y, hy, st, dtype=np.float64)
    o, dx, dw, t_after = _run(kind, x, w, y, hy, t0=0.123, lamb=st.lamb)
    ok = np.ones(ref.logits.shape, dtype=bool)
    if kind == H.CURR:
        # CurricularFace's hard-example mask cos > cos(theta_y + m) (criterion.py:560-565) is a step: of 11 M cosines a
        # handful sit within fp32 rounding of their row's threshold and take the other branch than float64 does (the
        # reference's own fp32 run has the same freedom).  They are left out of the logit comparison, and counted.
        ty = ref.cos_s[np.arange(N), y] / hy.s
        ctm = ty * np.cos(hy.m) - np.sqrt(np.maximum(0.0, 1 - ty * ty)) * np.sin(hy.m)
        ok = np.abs(ref.cos_s / hy.s - ctm[:, None]) > 3e-6
        assert (~ok).sum() <= max(2, 1e-5 * ok.size)
    np.testing.assert_allclose(o["logits"].cpu().numpy()[ok], ref.logits[ok], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(o["cos_s"].cpu().numpy(), ref.cos_s, atol=LOGIT_TOL, rtol=0)
    assert abs(o["loss"].item() - ref.loss) < 1e-3
    np.testing.assert_allclose(o["lse"].cpu().numpy(), ref.lse, atol=1e-3)
    assert tuple(o["topk"].cpu().numpy()) == (ref.top1, ref.top5)
    np.testing.assert_allclose(dx, ref.dx, atol=1e-3 * np.abs(ref.dx).max(), rtol=0)
    np.testing.assert_allclose(dw, ref.dw, atol=1e-3 * np.abs(ref.dw).max(), rtol=0)
    if kind == H.CURR:
        assert t_after == pytest.approx(st.t, abs=1e-6)


def test_pair_cosine_and_threshold(golden_dir):
    from frx import ops
    g = np.load(os.path.join(golden_dir, "verify_threshold.npz"))
    dev = torch.device("cuda:0")
    for tag in "abc":
        cos = ops.pair_cosine(torch.from_numpy(g[f"{tag}_f1"]).to(dev), torch.from_numpy(g[f"{tag}_f2"]).to(dev))
        np.testing.assert_allclose(cos.cpu().numpy(), g[f"{tag}_cos"], atol=2e-6)
        same = torch.from_numpy(g[f"{tag}_same"]).to(dev)
        # feed the reference's own similarities so the strict '>' at a tie is exercised bit-exactly
        cref = torch.from_numpy(g[f"{tag}_cos"]).to(dev)
        for thr, acc in zip(g[f"{tag}_eval_thr"], g[f"{tag}_eval_acc"]):
            cnt = int(ops.threshold_count(cref, same, float(np.float32(thr))).item())
            # the reference compares fp32 cos with a python float threshold promoted in fp32
            assert 100.0 * cnt / len(same) == pytest.approx(acc, abs=1e-9)
    # LFW size: 6000 pairs x 512, vs oracle; plus empty input
    rng = np.random.RandomState(0)
    f1, f2 = rng.randn(6000, 512).astype(np.float32), rng.randn(6000, 512).astype(np.float32)
    cos = ops.pair_cosine(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)).cpu().numpy()
    np.testing.assert_allclose(cos, V.pair_cosine(f1.astype(np.float64), f2.astype(np.float64), np.float64), atol=1e-6)
    assert ops.pair_cosine(torch.zeros(0, 512, device=dev), torch.zeros(0, 512, device=dev)).numel() == 0
