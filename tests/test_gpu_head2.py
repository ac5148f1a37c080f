"""GPU parity for the SURVEY 8(f)-3 heads (MV-Softmax am/arc, AdaFace, ElasticArcFace, ElasticCosFace, MagFace):
HIP forward + CE + loss_g + backward through the C ABI, against the reference's golden vectors
(tests/golden/make_golden_heads2.py) and against the float64 oracle on seeded / ragged shapes."""
import os

import zlib

import numpy as np
import pytest
import torch

from oracle import heads as H
from test_oracle_heads2 import CASES, ill_rows, load_case

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3          # north-star tolerance on logits (cosine x s)


def kind_params(kind, hyper):
    """-> (p, flags) of frx_head_desc for this kind"""
    if kind in (H.MV_AM, H.MV_ARC):
        return (hyper.mv_weight,), 0
    if kind == H.ADA:
        return (hyper.h, hyper.t_alpha), 0
    if kind == H.MAG:
        return (hyper.l_margin, hyper.u_margin, hyper.l_a, hyper.u_a), int(hyper.easy_margin)
    if kind == H.VPL:
        return (hyper.lamda, hyper.delta), int(hyper.easy_margin) | (2 if hyper.memory_on else 0)
    return (), 0


def run(kind, x, w, y, hyper, state, margins=None, lambda_g=0.0, dlogits_mode=False):
    from frx import ops
    dev = torch.device("cuda:0")
    N, D = x.shape
    Cc = w.shape[0] if H.weight_is_cd(kind) else w.shape[1]
    p, flags = kind_params(kind, hyper)
    ctx = ops.HeadContext(kind, N, D, Cc, hyper.s, float(hyper.m), device=dev, p=p, flags=flags, lambda_g=lambda_g)
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    wd = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dev)
    yd = torch.from_numpy(np.asarray(y).astype(np.int64)).to(dev)
    st = None
    if kind == H.ADA:
        st = torch.tensor([state.batch_mean, state.batch_std], dtype=torch.float32, device=dev)
    elif kind in (H.ELASTIC_ARC, H.ELASTIC_COS):
        st = torch.from_numpy(np.asarray(margins, dtype=np.float32)).to(dev)
    elif kind == H.VPL:                      # [mem | life]
        mem = np.zeros((Cc, D), np.float32) if state.mem is None else state.mem
        life = np.zeros(Cc, np.float32) if state.life is None else state.life
        st = torch.from_numpy(np.concatenate([mem.reshape(-1), life]).astype(np.float32)).to(dev)
    o = ops.head_forward(ctx, xd, wd, yd, state_t=st, want_logits=True, elastic_plus=hyper.plus)
    if dlogits_mode:       # autograd-style: upstream gradient of the mean CE w.r.t. the returned logits
        z = o["logits"].double()
        gz = (torch.softmax(z, 1) - torch.nn.functional.one_hot(yd, Cc)) / N
        dx, dw = ops.head_backward_dlogits(ctx, xd, wd, yd, gz.float().contiguous(), state_t=st)
    else:
        dx, dw = ops.head_backward(ctx, xd, wd, yd, state_t=st)
    torch.cuda.synchronize()
    return o, dx.cpu().numpy(), dw.cpu().numpy(), (None if st is None else st.cpu().numpy())


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("tag", ["fresh", "warm"])
def test_head_vs_reference_golden(golden_dir, name, tag):
    g, kind, hyper, st, margins = load_case(golden_dir, name, tag)
    lam = float(g["lambda_g"])
    o, dx, dw, st_after = run(kind, g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_y"], hyper, st, margins, lambda_g=lam)
    y = g[f"{tag}_y"]
    ill = ill_rows(g, tag, kind, hyper.s)
    ok = ~ill
    np.testing.assert_allclose(o["cos_s"].cpu().numpy(), g[f"{tag}_cos_s"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(o["logits"].cpu().numpy()[ok], g[f"{tag}_logits"][ok], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(o["logits"].cpu().numpy()[ill], g[f"{tag}_logits"][ill], atol=0.2, rtol=0)
    np.testing.assert_allclose(o["norms"].cpu().numpy(), g[f"{tag}_norms"].reshape(-1), rtol=1e-5)
    assert abs(o["loss"].item() - float(g[f"{tag}_loss"])) < (1e-3 if ok.all() else 2e-2)
    if "loss_g" in o:
        assert o["loss_g"].item() == pytest.approx(float(g[f"{tag}_loss_g"]), rel=1e-5, abs=1e-9)
    n = len(y)
    top = o["topk"].cpu().numpy()
    assert 100.0 * top[0] / n == pytest.approx(float(g[f"{tag}_acc1"]), abs=1e-4)
    assert 100.0 * top[1] / n == pytest.approx(float(g[f"{tag}_acc5"]), abs=1e-4)
    sx, sw = np.abs(g[f"{tag}_dx"]).max(), np.abs(g[f"{tag}_dw"]).max()
    np.testing.assert_allclose(dx[ok], g[f"{tag}_dx"][ok], atol=1e-3 * sx, rtol=0)
    wc = (lambda a: a) if H.weight_is_cd(kind) else (lambda a: a.T)
    okc = np.ones(wc(dw).shape[0], dtype=bool)
    okc[y[ill]] = False
    np.testing.assert_allclose(wc(dw)[okc], wc(g[f"{tag}_dw"])[okc], atol=1e-3 * sw, rtol=0)
    if kind == H.ADA:
        assert st_after[0] == pytest.approx(float(g[f"{tag}_post_batch_mean"]), rel=1e-5)
        assert st_after[1] == pytest.approx(float(g[f"{tag}_post_batch_std"]), rel=1e-5)
    if kind == H.VPL:
        Cc, D = g[f"{tag}_w"].shape
        np.testing.assert_allclose(st_after[:Cc * D].reshape(Cc, D), g[f"{tag}_post_mem"], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(st_after[Cc * D:], g[f"{tag}_post_life"])


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("shape", [(32, 512, 100), (48, 512, 1000), (7, 64, 37)])
def test_head_vs_oracle_seeded(name, shape):
    """Config-1 head shape (N=32, C=100) plus ragged shapes (odd C, N not a tile multiple); float64 oracle."""
    kind = CASES[name]
    N, D, Cc = shape
    rng = np.random.RandomState(zlib.crc32(repr((name, shape)).encode()) % 2**31)      # (hash() of a str is salted per process)
    wshape = (Cc, D) if H.weight_is_cd(kind) else (D, Cc)
    w = (rng.randn(*wshape) * 0.05).astype(np.float32)
    y = rng.randint(0, Cc, N)
    x = rng.randn(N, D).astype(np.float32)
    wc = w if H.weight_is_cd(kind) else w.T
    for i in range(0, N, 3):            # every third row sits near its class centre, norms spread over 4 .. 150
        x[i] = wc[y[i]] / np.linalg.norm(wc[y[i]]) * (4 + 146.0 * i / N) + 0.3 * rng.randn(D)
    hy = H.HeadHyper.default(kind)
    hy.easy_margin = name == "magface_easy"
    hy.plus = name.endswith("_plus")
    margins = None
    if kind in (H.ELASTIC_ARC, H.ELASTIC_COS):
        margins = np.clip(rng.normal(hy.m, 0.0125, N), hy.m - 0.0125, hy.m + 0.0125).astype(np.float32)
    def fresh_state():
        st = H.HeadState(batch_mean=21.5, batch_std=9.0)
        if kind == H.VPL:                   # a warm memory: half of the classes alive with random centres
            r2 = np.random.RandomState(7)
            st.mem = r2.randn(Cc, D).astype(np.float32)
            st.life = np.where(r2.rand(Cc) < 0.5, 5.0, -3.0).astype(np.float32)
        return st
    st = fresh_state()
    ref = H.head_forward_backward(kind, x, w, y, hy, fresh_state(), dtype=np.float64,
                                  row_margin=margins, lambda_g=20.0)
    o, dx, dw, _ = run(kind, x, w, y, hy, st, margins, lambda_g=20.0)
    np.testing.assert_allclose(o["logits"].cpu().numpy(), ref.logits, atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(o["cos_s"].cpu().numpy(), ref.cos_s, atol=LOGIT_TOL, rtol=0)
    assert abs(o["loss"].item() - ref.loss) < 1e-3
    assert tuple(o["topk"].cpu().numpy()) == (ref.top1, ref.top5)
    if "row_param" in ref.extra:
        np.testing.assert_allclose(o["row_param"].cpu().numpy(), ref.extra["row_param"], atol=2e-5)
    np.testing.assert_allclose(dx, ref.dx, atol=1e-3 * np.abs(ref.dx).max(), rtol=0)
    np.testing.assert_allclose(dw, ref.dw, atol=1e-3 * np.abs(ref.dw).max(), rtol=0)
    # the autograd-style entry (arbitrary dL/dlogits) gives the CE part of the same gradient
    ref0 = H.head_forward_backward(kind, x, w, y, hy, fresh_state(), dtype=np.float64, row_margin=margins, lambda_g=0.0)
    _, dx2, dw2, _ = run(kind, x, w, y, hy, fresh_state(), margins, dlogits_mode=True)
    np.testing.assert_allclose(dx2, ref0.dx, atol=1e-3 * np.abs(ref0.dx).max(), rtol=0)
    np.testing.assert_allclose(dw2, ref0.dw, atol=1e-3 * np.abs(ref0.dw).max(), rtol=0)


def test_state_is_required():
    from frx import ops
    from frx._lib import FrxError
    dev = torch.device("cuda:0")
    ctx = ops.HeadContext(H.ELASTIC_ARC, 8, 64, 10, 64.0, 0.5, device=dev)
    x, w = torch.randn(8, 64, device=dev), torch.randn(64, 10, device=dev)
    y = torch.zeros(8, dtype=torch.int64, device=dev)
    with pytest.raises(FrxError, match="per-row margins"):
        ops.head_forward(ctx, x, w, y, state_t=None)
    with pytest.raises(FrxError):
        ops.HeadContext(H.MAG, 8, 64, 10, 64.0, 0.0, device=dev, p=(0.45, 0.8, 110.0, 10.0))   # l_a > u_a


@pytest.mark.parametrize("name", ["mv_am", "mv_arc", "adaface", "elastic_arc", "elastic_cos", "magface", "vpl_arcface"])
def test_head_full_size_properties(name):
    """BASELINE head size (N=256, C=10575, D=512): size-independent properties instead of an O(N*C*D) oracle run --
    lse is the log-sum-exp of the returned logits, loss == mean(lse - z_y), non-target logits are s*cos except where the
    head re-weights them (MV / VPL), sampled cosines equal a float64 dot product, top-k equals torch.topk, gradients
    finite; for the scale-invariant heads dx is orthogonal to x."""
    from frx import ops
    kind = CASES[name]
    N, D, Cc = 256, 512, 10575
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    hy = H.HeadHyper.default(kind)
    wshape = (Cc, D) if H.weight_is_cd(kind) else (D, Cc)
    w = torch.randn(*wshape, generator=g) * 0.05
    x = torch.randn(N, D, generator=g)
    y = torch.randint(0, Cc, (N,), generator=g)
    p, flags = kind_params(kind, hy)
    ctx = ops.HeadContext(kind, N, D, Cc, hy.s, float(hy.m), device=dev, p=p, flags=flags, lambda_g=0.0)
    st = None
    if kind == H.ADA:
        st = torch.tensor([20.0, 100.0], device=dev)
    elif kind in (H.ELASTIC_ARC, H.ELASTIC_COS):
        st = torch.empty(N, device=dev).normal_(hy.m, 0.0125).clamp_(hy.m - 0.0125, hy.m + 0.0125)
    elif kind == H.VPL:
        st = torch.zeros(Cc * D + Cc, device=dev)
    xd, wd, yd = x.to(dev), w.to(dev), y.to(dev)
    o = ops.head_forward(ctx, xd, wd, yd, state_t=st, want_logits=True)
    z = o["logits"].double().cpu()
    lse = torch.logsumexp(z, dim=1)
    np.testing.assert_allclose(o["lse"].cpu().numpy(), lse.numpy(), atol=1e-3)
    assert abs(o["loss"].item() - (lse - z[torch.arange(N), y]).mean().item()) < 1e-3
    cs = o["cos_s"].cpu()
    if kind not in (H.MV_AM, H.MV_ARC, H.VPL):
        nt = torch.ones(N, Cc, dtype=torch.bool)
        nt[torch.arange(N), y] = False
        assert (o["logits"].cpu()[nt] - cs[nt]).abs().max().item() < 1e-3
    if kind != H.VPL:           # (VPL's pre-margin cosine is the memory blend, not the plain dot product)
        wc = (w if H.weight_is_cd(kind) else w.t()).double()
        xn = torch.nn.functional.normalize(x.double(), dim=1)
        idx = torch.randint(0, Cc, (64,), generator=g)
        cos = xn @ torch.nn.functional.normalize(wc[idx], dim=1).t()
        np.testing.assert_allclose((cs.double()[:, idx] / hy.s).numpy(), cos.clamp(-1, 1).numpy(), atol=1e-3 / 32)
    _, pred = cs.topk(5, 1, True, True)
    hit = pred.eq(y.view(-1, 1))
    assert int(hit[:, :1].sum()) == int(o["topk"][0]) and int(hit.sum()) == int(o["topk"][1])
    dx, dw = ops.head_backward(ctx, xd, wd, yd, state_t=st)
    assert torch.isfinite(dx).all() and torch.isfinite(dw).all() and dx.abs().max().item() > 0
    if kind in (H.MV_AM, H.MV_ARC, H.ELASTIC_ARC, H.ELASTIC_COS, H.VPL):      # scale-invariant in x: d/ds L(s*x) = 0
        rad = (dx.cpu() * x).sum(1).abs().max().item()
        assert rad < 1e-4 * dx.abs().max().item() * x.norm(dim=1).max().item() + 1e-6
    if kind == H.VPL:
        life = st[Cc * D:].cpu()
        assert int((life > 0).sum()) == len(set(y.tolist())) and float(life.max()) == hy.delta - 1
