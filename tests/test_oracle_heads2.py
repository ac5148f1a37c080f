"""Pin the oracle's restatement of the SURVEY 8(f)-3 heads (MV-Softmax am/arc, AdaFace, ElasticArcFace,
ElasticCosFace, MagFace with and without easy_margin) to golden vectors captured from the reference import
(tests/golden/make_golden_heads2.py): forward, CE, loss_g, analytic backward of CE + lambda_g * loss_g, state.  CPU only."""
import os

import numpy as np
import pytest

from oracle import heads as H

CASES = {"mv_am": H.MV_AM, "mv_arc": H.MV_ARC, "adaface": H.ADA, "elastic_arc": H.ELASTIC_ARC,
         "elastic_cos": H.ELASTIC_COS, "magface": H.MAG, "magface_easy": H.MAG, "vpl_arcface": H.VPL,
         "elastic_arc_plus": H.ELASTIC_ARC, "elastic_cos_plus": H.ELASTIC_COS}


def load_case(golden_dir, name, tag):
    g = np.load(os.path.join(golden_dir, f"heads_{name}.npz"))
    kind = CASES[name]
    hyper = H.HeadHyper.default(kind)
    if name == "magface_easy":
        hyper.easy_margin = True
    hyper.plus = name.endswith("_plus")
    st = H.HeadState(batch_mean=float(g[f"{tag}_pre_batch_mean"]), batch_std=float(g[f"{tag}_pre_batch_std"]))
    if kind == H.VPL:
        st.mem, st.life = g[f"{tag}_pre_mem"].copy(), g[f"{tag}_pre_life"].copy()
    margins = g[f"{tag}_row_margin"] if f"{tag}_row_margin" in g.files else None
    return g, kind, hyper, st, margins


def ill_rows(g, tag, kind, s):
    """Rows whose target cosine sits on a clamp limit: acos / sqrt(1 - c^2) there swing by 1e-3 per ulp of c in
    the reference itself (the fixture plants two such rows on purpose); they get a loose bound."""
    y = g[f"{tag}_y"]
    ty = g[f"{tag}_cos_s"][np.arange(len(y)), y] / s
    lim = 1e-3 if kind == H.ADA else 1e-5
    return np.abs(ty) > 1 - 2 * lim


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("tag", ["fresh", "warm"])
def test_head_matches_reference(golden_dir, name, tag):
    g, kind, hyper, st, margins = load_case(golden_dir, name, tag)
    lam = float(g["lambda_g"])
    out = H.head_forward_backward(kind, g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_y"], hyper, st, dtype=np.float32,
                                  row_margin=margins, lambda_g=lam)
    y = g[f"{tag}_y"]
    ill = ill_rows(g, tag, kind, hyper.s)
    ok = ~ill
    np.testing.assert_allclose(out.cos_s, g[f"{tag}_cos_s"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(out.logits[ok], g[f"{tag}_logits"][ok], atol=3e-4, rtol=0)
    np.testing.assert_allclose(out.logits[ill], g[f"{tag}_logits"][ill], atol=0.2, rtol=0)
    np.testing.assert_allclose(out.norms, g[f"{tag}_norms"], rtol=1e-6)           # MagFace: the clamped x_norm
    assert abs(out.loss - float(g[f"{tag}_loss"])) < (1e-4 if not ill.any() else 2e-2)
    assert out.extra.get("loss_g", 0.0) == pytest.approx(float(g[f"{tag}_loss_g"]), rel=1e-5, abs=1e-9)
    sx, sw = np.abs(g[f"{tag}_dx"]).max(), np.abs(g[f"{tag}_dw"]).max()
    np.testing.assert_allclose(out.dx[ok], g[f"{tag}_dx"][ok], atol=3e-4 * sx, rtol=0)
    wc = (lambda a: a) if H.weight_is_cd(kind) else (lambda a: a.T)
    okc = np.ones(wc(out.dw).shape[0], dtype=bool)
    okc[y[ill]] = False
    np.testing.assert_allclose(wc(out.dw)[okc], wc(g[f"{tag}_dw"])[okc], atol=3e-4 * sw, rtol=0)
    n = len(y)
    assert 100.0 * out.top1 / n == pytest.approx(float(g[f"{tag}_acc1"]), abs=1e-4)
    assert 100.0 * out.top5 / n == pytest.approx(float(g[f"{tag}_acc5"]), abs=1e-4)
    if kind == H.ADA:
        assert st.batch_mean == pytest.approx(float(g[f"{tag}_post_batch_mean"]), rel=1e-5)
        assert st.batch_std == pytest.approx(float(g[f"{tag}_post_batch_std"]), rel=1e-5)
    if kind == H.VPL:
        np.testing.assert_allclose(st.mem, g[f"{tag}_post_mem"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(st.life, g[f"{tag}_post_life"])
    assert float(g[f"{tag}_onehot_sum"]) == n


def test_fixture_exercises_the_branches(golden_dir):
    """MV mask fires on non-targets; MagFace sees norms below l_a, inside, and above u_a, and both margin branches;
    AdaFace's margin scaler takes both signs; the elastic margins are inside [m - std, m + std]."""
    g = np.load(os.path.join(golden_dir, "heads_mv_arc.npz"))
    y = g["fresh_y"]; rows = np.arange(len(y))
    c = g["fresh_cos_s"] / 32.0
    moved = ~np.isclose(g["fresh_logits"], g["fresh_cos_s"], atol=1e-6)
    moved[rows, y] = False
    assert moved.any()
    g = np.load(os.path.join(golden_dir, "heads_magface.npz"))
    nr = np.linalg.norm(g["fresh_x"], axis=1)
    assert (nr < 10).any() and (nr > 110).any() and ((nr > 10) & (nr < 110)).any()
    y = g["fresh_y"]; rows = np.arange(len(y))
    ty = g["fresh_cos_s"][rows, y] / 64.0
    am = (0.8 - 0.45) / 100.0 * (np.clip(nr, 10, 110) - 10) + 0.45
    assert (ty > np.cos(np.pi - am)).any() and (ty <= np.cos(np.pi - am)).any()
    g, kind, hyper, st, _ = load_case(golden_dir, "adaface", "warm")
    out = H.head_forward_backward(kind, g["warm_x"], g["warm_w"], g["warm_y"], hyper, st, need_grad=False)
    assert (out.extra["row_param"] > 0).any() and (out.extra["row_param"] < 0).any()
    g = np.load(os.path.join(golden_dir, "heads_elastic_arc.npz"))
    assert np.all(np.abs(g["fresh_row_margin"] - 0.5) <= 0.0125 + 1e-7) and g["fresh_row_margin"].std() > 0
    # VPL: in the warm state classes of EARLIER batches are still alive, so the memory path acts on non-targets,
    # and some classes have never been seen (inactive columns)
    g = np.load(os.path.join(golden_dir, "heads_vpl_arcface.npz"))
    alive_before = g["warm_pre_life"] > 1
    assert alive_before.any() and not set(np.where(alive_before)[0]) <= set(g["warm_y"].tolist())
    assert (g["warm_post_life"] <= 0).any() and len(set(g["warm_y"].tolist())) < len(g["warm_y"])


@pytest.mark.parametrize("name", list(CASES))
def test_float64_gradients_tight(golden_dir, name):
    g, kind, hyper, st, margins = load_case(golden_dir, name, "warm")
    out = H.head_forward_backward(kind, g["warm_x"], g["warm_w"], g["warm_y"], hyper, st, dtype=np.float64,
                                  row_margin=margins, lambda_g=float(g["lambda_g"]))
    ok = ~ill_rows(g, "warm", kind, hyper.s)
    np.testing.assert_allclose(out.dx[ok], g["warm_dx"][ok], atol=3e-4 * np.abs(g["warm_dx"]).max(), rtol=0)


def test_torch_head_matches_closed_form():
    """oracle/resnet50.TorchHead (autograd, drives the whole-step oracle) == oracle/heads closed form for the new kinds,
    including MagFace's loss_g term and the elastic heads' injected margins."""
    import torch
    import torch.nn.functional as F
    from oracle.resnet50 import TorchHead
    rng = np.random.RandomState(0)
    for kind in (H.MV_AM, H.MV_ARC, H.ADA, H.ELASTIC_ARC, H.ELASTIC_COS, H.MAG, H.VPL):
        torch.manual_seed(kind)
        hy = H.HeadHyper.default(kind)
        th = TorchHead(kind, 64, 40, hy)
        x = rng.randn(16, 64).astype(np.float32)
        x[::3] *= 3
        x = torch.from_numpy(x).requires_grad_(True)
        y = torch.from_numpy(rng.randint(0, 40, 16))
        cos_s, logits = th(x, y)
        loss_id = F.cross_entropy(logits, y)
        (loss_id + 7.0 * th.loss_g).backward()
        out = H.head_forward_backward(kind, x.detach().numpy(), th.weight.detach().numpy(), y.numpy(), hy, H.HeadState(),
                                      dtype=np.float32, lambda_g=7.0,
                                      row_margin=None if th.row_margin is None else th.row_margin.numpy())
        assert abs(out.loss - loss_id.item()) < 1e-4
        np.testing.assert_allclose(out.logits, logits.detach().numpy(), atol=2e-4)
        np.testing.assert_allclose(out.dx, x.grad.numpy(), atol=2e-4 * np.abs(out.dx).max())
        np.testing.assert_allclose(out.dw, th.weight.grad.numpy(), atol=2e-4 * np.abs(out.dw).max())
