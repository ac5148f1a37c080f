"""Pin the oracle (oracle/heads.py) to the golden vectors captured from the reference
import (tests/golden/make_golden.py): forward, CE, analytic backward, state updates,
top-k, CustomStepLR.  CPU only."""
import os

import numpy as np
import pytest

from oracle import heads as H

KINDS = {"arcface": H.ARC, "cosface": H.COS, "sphereface": H.SPHERE, "curricular": H.CURR,
         "arcface_easy": H.ARC, "sphereface_m4": H.SPHERE}      # (two more constructor variants of the same heads)


def hyper_for(name):
    """the reference constructor arguments behind each fixture (tests/golden/make_golden.py)"""
    hy = H.HeadHyper.default(KINDS[name])
    if name == "arcface_easy":
        hy.easy_margin = True                    # ArcFace(easy_margin=True): criterion.py:284-285
    if name == "sphereface_m4":
        hy.m = 4                                 # SphereFace's own default m (criterion.py:17); config.py:17 ships 2
    return hy


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"heads_{name}.npz"))


@pytest.mark.parametrize("name", list(KINDS))
@pytest.mark.parametrize("tag", ["fresh", "warm"])
def test_head_matches_reference(golden_dir, name, tag):
    g = _load(golden_dir, name)
    kind = KINDS[name]
    st = H.HeadState(iter=int(g[f"{tag}_pre_iter"]), t=float(g[f"{tag}_pre_t"]))
    out = H.head_forward_backward(kind, g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_y"],
                                  hyper_for(name), st, dtype=np.float32)
    # logits are cosines x 64: the north-star bar is 1e-3; the fp32 restatement sits far below it,
    # EXCEPT where the reference itself is ill-conditioned: a target cosine within ~1e-5 of +-1
    # makes sqrt(1-c^2) (ArcFace :281, Curricular :555) swing by 3e-4 per ulp of c.  Those rows
    # (the fixture plants one on purpose) are held to a loose bound instead.
    y = g[f"{tag}_y"]
    rows = np.arange(len(y))
    ty = g[f"{tag}_cos_s"][rows, y] / (64.0 if kind != H.SPHERE else g[f"{tag}_norms"][:, 0])
    ill = (np.abs(ty) > 1 - 1e-5) & (kind in (H.ARC, H.CURR))
    ok = ~ill
    np.testing.assert_allclose(out.cos_s, g[f"{tag}_cos_s"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(out.logits[ok], g[f"{tag}_logits"][ok], atol=2e-4, rtol=0)
    np.testing.assert_allclose(out.logits[ill], g[f"{tag}_logits"][ill], atol=5e-2, rtol=0)
    np.testing.assert_allclose(out.norms, g[f"{tag}_norms"], rtol=1e-6)
    assert abs(out.loss - float(g[f"{tag}_loss"])) < (1e-4 if not ill.any() else 5e-3)
    sx = np.nanmax(np.abs(g[f"{tag}_dx"]))
    sw = np.nanmax(np.abs(g[f"{tag}_dw"]))
    np.testing.assert_allclose(out.dx[ok], g[f"{tag}_dx"][ok], atol=2e-4 * sx, rtol=0)
    wc = (lambda a: a) if H.weight_is_cd(kind) else (lambda a: a.T)
    okc = np.ones(wc(out.dw).shape[0], dtype=bool)
    okc[y[ill]] = False
    np.testing.assert_allclose(wc(out.dw)[okc], wc(g[f"{tag}_dw"])[okc], atol=2e-4 * sw, rtol=0)
    n = len(g[f"{tag}_y"])
    assert 100.0 * out.top1 / n == pytest.approx(float(g[f"{tag}_acc1"]), abs=1e-4)
    assert 100.0 * out.top5 / n == pytest.approx(float(g[f"{tag}_acc5"]), abs=1e-4)
    assert st.iter == int(g[f"{tag}_post_iter"]) or kind != H.SPHERE
    if kind == H.SPHERE:
        assert st.lamb == pytest.approx(float(g[f"{tag}_lamb"]), rel=1e-12)
    if kind == H.CURR:
        assert st.t == pytest.approx(float(g[f"{tag}_post_t"]), abs=1e-6)
    assert float(g[f"{tag}_loss_g"]) == 0.0
    assert float(g[f"{tag}_onehot_sum"]) == n


@pytest.mark.parametrize("name", list(KINDS))
def test_head_float64_gradients_tight(golden_dir, name):
    """In float64 the closed form agrees with the (fp32) reference autograd to fp32 rounding."""
    g = _load(golden_dir, name)
    kind = KINDS[name]
    st = H.HeadState(iter=int(g["warm_pre_iter"]), t=float(g["warm_pre_t"]))
    out = H.head_forward_backward(kind, g["warm_x"], g["warm_w"], g["warm_y"],
                                  hyper_for(name), st, dtype=np.float64)
    y = g["warm_y"]
    ty = g["warm_cos_s"][np.arange(len(y)), y] / (64.0 if kind != H.SPHERE else g["warm_norms"][:, 0])
    ok = ~((np.abs(ty) > 1 - 1e-5) & (kind in (H.ARC, H.CURR)))   # see test_head_matches_reference
    assert abs(out.loss - float(g["warm_loss"])) < (1e-4 if ok.all() else 5e-3)
    np.testing.assert_allclose(out.dx[ok], g["warm_dx"][ok], atol=3e-4 * np.nanmax(np.abs(g["warm_dx"])), rtol=0)


def test_edge_rows_exercise_both_margin_branches(golden_dir):
    g = _load(golden_dir, "arcface")
    cos = g["fresh_cos_s"] / 64.0
    ty = cos[np.arange(len(g["fresh_y"])), g["fresh_y"]]
    th = np.cos(np.pi - 0.5)
    assert (ty > th).any() and (ty <= th).any()
    assert g["fresh_y"][0] == 0 and g["fresh_y"][1] == cos.shape[1] - 1
    gc = _load(golden_dir, "curricular")
    c = gc["fresh_cos_s"] / 64.0
    y = gc["fresh_y"]
    ty = c[np.arange(len(y)), y][:, None]
    cm = ty * np.cos(0.5) - np.sqrt(1 - ty * ty) * np.sin(0.5)
    mask = c > cm
    mask[np.arange(len(y)), y] = False
    assert mask.any(), "Curricular hard-negative mask must fire in the fixture"


def test_accuracy_topk_matches_counts(golden_dir):
    g = _load(golden_dir, "cosface")
    a1, a5 = H.accuracy_topk(g["fresh_cos_s"], g["fresh_y"], (1, 5))
    assert a1 == pytest.approx(float(g["fresh_acc1"]), abs=1e-4)
    assert a5 == pytest.approx(float(g["fresh_acc5"]), abs=1e-4)


def test_custom_step_lr(golden_dir):
    lrs = np.load(os.path.join(golden_dir, "customstep_lr.npz"))["lrs"]
    mine = H.custom_step_lr(0.1, 70)
    np.testing.assert_allclose(mine, lrs, rtol=1e-12)
    assert lrs[19] == pytest.approx(0.1) and lrs[20] == pytest.approx(0.01) and lrs[60] == pytest.approx(1e-4)


def test_torch_head_matches_closed_form():
    """oracle/resnet50.TorchHead (autograd) == oracle/heads closed form, all four kinds."""
    import torch
    import torch.nn.functional as F
    from oracle.resnet50 import TorchHead
    rng = np.random.RandomState(0)
    for kind in (H.ARC, H.COS, H.SPHERE, H.CURR):
        torch.manual_seed(kind)
        th = TorchHead(kind, 64, 40, H.HeadHyper.default(kind))
        x = torch.from_numpy(rng.randn(16, 64).astype(np.float32)).requires_grad_(True)
        y = torch.from_numpy(rng.randint(0, 40, 16))
        cos_s, logits = th(x, y)
        loss = F.cross_entropy(logits, y)
        loss.backward()
        out = H.head_forward_backward(kind, x.detach().numpy(), th.weight.detach().numpy(), y.numpy(),
                                      H.HeadHyper.default(kind), H.HeadState(), dtype=np.float32)
        assert abs(out.loss - loss.item()) < 1e-4
        np.testing.assert_allclose(out.logits, logits.detach().numpy(), atol=2e-4)
        np.testing.assert_allclose(out.dx, x.grad.numpy(), atol=2e-4 * np.abs(out.dx).max())
        np.testing.assert_allclose(out.dw, th.weight.grad.numpy(), atol=2e-4 * np.abs(out.dw).max())
