"""Properties of the training step at BASELINE configs[1] size (ArcFace R50, 10 575 classes, batch 256, bf16), where the
CPU oracle is too slow to run: determinism of the forward, linearity of the backward in the upstream gradient, the two
weight-gradient schedules (per-layer launches / one grouped launch) against each other, and a sane loss curve."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, C = 256, 10575


def _engine(grouped=True, seed=0):
    from frx import engine as E, ops
    os.environ["FRX_WGRAD_GROUPED"] = "1" if grouped else "0"
    try:
        return E.FaceEngine("arcface", C, N, dtype=ops.BF16, device=DEV, seed=seed)
    finally:
        os.environ.pop("FRX_WGRAD_GROUPED", None)


def _batch(seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).to(DEV), torch.randint(0, C, (N,), generator=g).to(DEV)


def test_forward_is_bit_reproducible_and_finite():
    eng = _engine()
    x, y = _batch()
    a = eng.forward_loss(x, y)
    f1, l1 = a["feats"].clone(), a["loss"].clone()
    b = eng.forward_loss(x, y)
    assert torch.equal(b["feats"], f1) and torch.equal(b["loss"], l1), "the forward has no atomics: it must replay bit for bit"
    assert torch.isfinite(f1).all() and 5.0 < l1.item() < 100.0
    for c in eng.net.convs:                      # train-mode BN really normalised every layer of this batch
        m, s = eng.net._bn(eng.net.bn_mean, c), eng.net._bn(eng.net.bn_invstd, c)
        assert torch.isfinite(m).all() and torch.isfinite(s).all() and (s > 0).all()


def test_backward_is_linear_in_the_upstream_gradient():
    """dW(2*g) == 2*dW(g) through 53 BatchNorm backward passes, fused prologues, masks and atomics (all linear in g)."""
    eng = _engine()
    net = eng.net
    x, y = _batch(1)
    eng.forward_loss(x, y)
    g = torch.Generator().manual_seed(5)
    df = (torch.randn(N, 512, generator=g) * 1e-3).to(DEV)
    net.zero_grad(); net.backward(df); g1 = net.grads.clone()
    net.zero_grad(); net.backward(df * 2); g2 = net.grads.clone()
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    rel = ((g2 - 2 * g1).norm() / (2 * g1).norm()).item()
    assert rel < 2e-2, rel                        # bf16 rounding of dz / dy is the only non-linear part
    for c in (net.stem, net.blocks[0].conv2, net.blocks[7].conv3, net.blocks[-1].conv1):
        a, b = net.w_grad(c, g1).flatten(), net.w_grad(c, g2).flatten()
        cos = torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)
        assert cos.item() > 0.999, (c.name, cos.item())


def test_grouped_and_per_layer_weight_gradients_agree():
    """Same upstream gradient into both schedules: the dgrad chain has no atomics, so every dy is bit-identical and the
    weight gradients may differ only by the order of their fp32 atomics.  (Through the head the comparison is useless:
    its atomics perturb dfeat by 1e-7, bf16 rounding flips amplify that to ~2e-2 by the stem -- measured.)"""
    e1, e2 = _engine(grouped=True), _engine(grouped=False)
    assert e1.net.grouped_wgrad and not e2.net.grouped_wgrad
    x, y = _batch(2)
    g = torch.Generator().manual_seed(9)
    df = (torch.randn(N, 512, generator=g) * 1e-3).to(DEV)
    grads = []
    for e in (e1, e2):
        e.net.training = True
        e.net.zero_grad()
        e.forward_loss(x, y)
        e.net.backward(df)
        grads.append(e.net.grads.clone())
    a, b = grads
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert torch.equal(e1.net.blocks[0].dz3, e2.net.blocks[0].dz3), "the input-gradient chain must not depend on the schedule"
    for c in e1.net.convs:
        ga, gb = e1.net.w_grad(c, a), e1.net.w_grad(c, b)
        r = ((ga - gb).norm() / (gb.norm() + 1e-30)).item()
        assert r < 1e-4, (c.name, r)


def test_loss_goes_down_on_a_repeated_batch():
    eng = _engine()
    x, y = _batch(3)
    ls = [eng.train_step(x, y, 0.005)["loss"].item() for _ in range(30)]
    assert all(np.isfinite(ls)), ls
    assert max(ls[-4:]) < ls[0] - 3.0, ls


def test_upper_gradients_are_final_after_the_first_backward_half():
    """What the overlapped all-reduce relies on: after step_upper() the upper gradient ranges (layer3/4, fc, head: 94 %
    of the bytes) do not change any more, and the lower ranges are still untouched."""
    eng = _engine()
    x, y = _batch(4)
    eng.step_upper(x, y)
    g1 = eng.net.grads.clone()
    rng = eng.net.grad_ranges()
    for lo, hi in rng["lower"]:
        assert g1[lo:hi].abs().max().item() == 0.0, "a lower-range gradient was written before the lower half ran"
    for lo, hi in rng["upper"]:
        assert g1[lo:hi].abs().max().item() > 0.0
    up = sum(hi - lo for lo, hi in rng["upper"])
    assert up / eng.net.n_params > 0.9
    eng.step_lower()
    g2 = eng.net.grads
    for lo, hi in rng["upper"]:
        assert torch.equal(g2[lo:hi], g1[lo:hi])
    for c in eng.net.convs:
        assert eng.net.w_grad(c).abs().max().item() > 0.0, c.name
    assert torch.isfinite(g2).all()


def test_graph_replay_stays_finite_unsynchronised():
    """Regression guard for the replay-only failures of DESIGN.md section 8 (a memset node that filled with a stale
    pattern; a plan table wiped after its copy): 40 back-to-back replays of the captured full-size step, no host
    synchronisation in between, must keep the loss finite, the parameters small and every conv gradient non-zero."""
    eng = _engine()
    x, y = _batch(6)
    eng.net.lr_dev.fill_(0.005)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.train_step(x, y)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = eng.train_step(x, y)
    trace = torch.zeros(40, device=DEV)
    for i in range(40):
        graph.replay()
        trace[i].copy_(out["loss"].reshape(()))
    torch.cuda.synchronize()
    ls = trace.tolist()
    assert all(np.isfinite(ls)), ls
    assert ls[-1] < ls[0], ls                     # the same batch every replay: the loss must come down
    assert eng.net.params.abs().max().item() < 50.0
    for c in eng.net.convs:
        g = eng.net.w_grad(c)
        assert torch.isfinite(g).all() and g.abs().max().item() > 0.0, c.name
