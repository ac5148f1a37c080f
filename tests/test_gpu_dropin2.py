"""GPU: the drop-in *Net wrappers of the SURVEY 8(f)-3 heads (MV_SoftmaxNet, AdaFaceNet, ElasticArcFaceNet,
ElasticCosFaceNet, MagFaceNet): reference forward contract, autograd path against the CPU oracle holding the same
weights (and, for the elastic heads, the same sampled margins), and the fused train_model path with lambda_g."""
import types

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import heads as H
from oracle.resnet50 import FaceNet
from test_gpu_dropin import DEV, LOGIT_TOL, _batch, _mk

pytestmark = pytest.mark.gpu
NETS = [("MV_SoftmaxNet", H.MV_AM, "mv_head.weight"), ("AdaFaceNet", H.ADA, "adaface.kernel"),
        ("ElasticArcFaceNet", H.ELASTIC_ARC, "head.kernel"), ("ElasticCosFaceNet", H.ELASTIC_COS, "head.kernel"),
        ("MagFaceNet", H.MAG, "magface.kernel"), ("VPLArcFaceNet", H.VPL, "vpl_head.weight")]
LAMBDA_G = 35.0


def _oracle_twin(m, kind, C, pname):
    ref = FaceNet(kind, C)
    sd = m.state_dict()
    assert pname in sd, sorted(k for k in sd if not k.startswith("backbone."))
    ref.backbone.load_state_dict({k[len("backbone."):]: v.cpu() for k, v in sd.items() if k.startswith("backbone.")})
    with torch.no_grad():
        ref.head.weight.copy_(sd[pname].cpu())
    return ref


@pytest.mark.parametrize("cls,kind,pname", NETS)
def test_forward_contract_and_autograd_path_vs_oracle(cls, kind, pname):
    N, C = 6, 40
    m = _mk(cls, C, "f32", seed=2)
    ref = _oracle_twin(m, kind, C, pname)
    if kind == H.ADA:
        assert {"adaface.t", "adaface.batch_mean", "adaface.batch_std"} <= set(m.state_dict())
    x, y = _batch(N, C, 3)
    m.train()
    (cos_s, logits), norms, loss_g, one_hot = m(x, y)
    assert cos_s.shape == (N, C) and logits.shape == (N, C) and norms.shape == (N, 1)
    assert one_hot.sum().item() == N and logits.requires_grad
    if kind == H.MAG:
        assert torch.is_tensor(loss_g) and loss_g.requires_grad
    else:
        assert loss_g == 0
    loss_id = nn.CrossEntropyLoss()(logits, y)
    loss = loss_id + LAMBDA_G * loss_g                              # model_utils.py:180
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    loss.backward()
    if kind in (H.ELASTIC_ARC, H.ELASTIC_COS):                      # the oracle replays the margins the device drew
        marg = m._primary.t.detach().cpu().clone()
        hm = m.head.m
        assert marg.shape == (N,) and (marg - hm).abs().max().item() <= 0.0125 + 1e-6 and marg.std().item() > 0
        ref.head.next_margin = marg
    ref.train()
    (rc, rl), rf = ref(x.cpu(), y.cpu())
    rloss_id = F.cross_entropy(rl, y.cpu())
    rloss = rloss_id + LAMBDA_G * ref.head.loss_g
    rloss.backward()
    assert abs(loss_id.item() - rloss_id.item()) < 1e-3
    assert (logits.detach().cpu() - rl.detach()).abs().max().item() < LOGIT_TOL
    assert (cos_s.cpu() - rc.detach()).abs().max().item() < LOGIT_TOL
    if kind == H.MAG:
        assert loss_g.detach().item() == pytest.approx(float(ref.head.loss_g.detach()), rel=1e-4)
        rn = rf.detach().norm(dim=1).clamp(10.0, 110.0)
        assert torch.allclose(norms.view(-1).cpu(), rn, rtol=1e-3)  # the CLAMPED norms (criterion.py:1291)
    if kind == H.ADA:
        assert m.head.batch_mean.item() == pytest.approx(ref.head.state.batch_mean, rel=1e-4)
        assert m.head.batch_std.item() == pytest.approx(ref.head.state.batch_std, rel=1e-4)
        assert m.state_dict()["adaface.batch_mean"].item() == pytest.approx(ref.head.state.batch_mean, rel=1e-4)
    g, rg = m.backbone.fc.weight.grad, ref.backbone.fc.weight.grad
    assert (g.cpu() - rg).norm().item() < 0.05 * rg.norm().item()
    hg, rhg = m.head._param().grad, ref.head.weight.grad
    assert (hg.cpu() - rhg).norm().item() < 0.02 * rhg.norm().item()
    if kind == H.VPL:
        # the class memory carries over: a second batch sees the first batch's classes as live non-target proxies
        assert {"vpl_head.mem", "vpl_head.life", "vpl_head.cos_m", "vpl_head.th"} <= set(m.state_dict())
        assert (m.head.life > 0).sum().item() == len(set(y.tolist()))
        x2, y2 = _batch(N, C, 4)
        (c2, l2), _, _, _ = m(x2, y2)
        (rc2, rl2), _ = ref(x2.cpu(), y2.cpu())
        assert (l2.detach().cpu() - rl2.detach()).abs().max().item() < LOGIT_TOL
        assert torch.allclose(m.head.life.cpu(), ref.head.state.life)
        assert (m.head.mem.cpu() - ref.head.state.mem).abs().max().item() < 1e-3 * ref.head.state.mem.abs().max().item()
        # memory switched off (change_training_mode(False), criterion.py:676): plain ArcFace-style logits, memory untouched
        m.change_training_mode(False)
        ref.head.hyper.memory_on = False
        life_before = m.head.life.clone()
        (c3, l3), _, _, _ = m(x2, y2)
        (rc3, rl3), _ = ref(x2.cpu(), y2.cpu())
        assert (l3.detach().cpu() - rl3.detach()).abs().max().item() < LOGIT_TOL
        assert torch.equal(life_before, m.head.life)


def test_magface_fused_train_model_uses_lambda_g():
    """fused path (whole step in the engine) == autograd-compatible path, with loss = loss_id + lambda_g * loss_g"""
    from utils import model_utils as MU
    C = 30
    m = _mk("MagFaceNet", C, "f32", seed=4)
    m2 = _mk("MagFaceNet", C, "f32", seed=5)
    m2.load_state_dict(m.state_dict())
    crit = nn.CrossEntropyLoss().to(DEV)
    args = types.SimpleNamespace(lambda_g=LAMBDA_G, print_freq=1)
    data = [tuple(t.cpu() for t in _batch(8, C, 10 + i)) for i in range(2)]
    opt = MU.make_optimizer(m, 0.01)
    loss_fused = MU.train_model(m, data, crit, opt, MU.GradScaler(enabled=False), DEV, 1, 1, args)
    opt2 = torch.optim.SGD(m2.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    loss_compat = MU.train_model(m2, data, crit, opt2, MU.GradScaler(enabled=False), DEV, 1, 1, args)
    assert np.isfinite(loss_fused) and loss_fused == pytest.approx(loss_compat, rel=2e-3)
    # loss_g contributes: LAMBDA_G * mean(x_norm / u_a^2 + 1 / x_norm) >= LAMBDA_G * 2 / u_a
    assert loss_fused > LAMBDA_G * 2 / 110.0
    w1, w2 = m.backbone.fc.weight.detach(), m2.backbone.fc.weight.detach()
    assert (w1 - w2).norm().item() < 1e-3 * w1.norm().item()


@pytest.mark.parametrize("head", ["mv_arc", "adaface", "elastic_arc", "magface", "vpl_arcface"])
def test_engine_bf16_fused_steps_run(head):
    """bf16 speed mode: fused steps stay finite, move the weights, refresh the elastic margins / AdaFace statistics.
    (Gradient parity is pinned above; on a 16-image batch the margin losses first rise for every head, ArcFace included.)"""
    from frx import engine as E
    eng = E.FaceEngine(head, 64, 16, dtype=E.BF16, device=DEV, seed=0, lambda_g=LAMBDA_G if head == "magface" else 0.0)
    x, y = _batch(16, 64, 1)
    w0, st0 = eng.head_w().clone(), eng.t.clone()
    for i in range(6):
        out = eng.train_step(x, y, 0.002)
        assert np.isfinite(out["loss"].item())
        if head == "magface":
            assert 2 / 110.0 <= out["loss_g"].item() < 1.0
    assert not torch.equal(w0, eng.head_w())
    if head in ("elastic_arc", "adaface", "vpl_arcface"):
        assert not torch.equal(st0, eng.t)


def test_elastic_margins_are_redrawn_on_every_graph_replay():
    """The elastic heads draw their margins inside the captured step (criterion.py:1002 draws them inside forward): every
    replay of the hipGraph must see fresh ones (torch's graph-safe Philox offsets), not the capture-time draw."""
    from frx import engine as E
    eng = E.FaceEngine("elastic_arc", 64, 16, dtype=E.BF16, device=DEV, seed=0)
    x, y = _batch(16, 64, 1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.train_step(x, y, 0.002)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.train_step(x, y, 0.002)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        m = eng.t.clone()
        assert (m - 0.5).abs().max().item() <= 0.0125 + 1e-6 and np.isfinite(out["loss"].item())
        seen.append(m)
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def _rank_matched(marg, ty):
    """criterion.py:1006-1011 as written: margin = sort(margin)[argsort(target_cos, descending)]"""
    return torch.sort(marg).values[torch.sort(ty, descending=True).indices]


@pytest.mark.parametrize("cls", ["ElasticArcFaceNet", "ElasticCosFaceNet"])
def test_elastic_plus_rank_matches_the_margins(cls):
    """plus=True: the margins the loss phase applied are the sorted draw indexed by the rank of the target cosines, on
    the autograd path, in the fused step, and in a replayed hipGraph of it (the sort runs inside the captured step)."""
    from utils import model_utils as MU
    N, C = 12, 40
    m = _mk(cls, C, "f32", seed=6)
    m.head.plus = True
    x, y = _batch(N, C, 8)
    m.train()
    (cos_s, logits), _, _, _ = m(x, y)
    eng = m._primary
    assert eng.elastic_plus
    ty = (cos_s / m.head.s)[torch.arange(N), y]
    marg = eng.t.clone()
    assert torch.equal(marg, _rank_matched(marg, ty)) and marg.std().item() > 0
    zt = logits.detach()[torch.arange(N), y] / m.head.s
    want = torch.cos(torch.acos(ty) + marg) if cls == "ElasticArcFaceNet" else ty - marg
    assert (zt - want).abs().max().item() < LOGIT_TOL / m.head.s * 4
    # fused step, eager then replayed
    opt = MU.make_optimizer(m, 0.001)
    st = m._stepper_for(eng, x)
    for i in range(3):
        out = st.step(x, y, 0.001)
        torch.cuda.synchronize()
        assert np.isfinite(out["loss"].item())
        marg = eng.t.clone()
        assert (marg - m.head.m).abs().max().item() <= 0.0125 + 1e-6
        ty_step = eng.margin_scratch.clone()             # the target cosines this step ranked by
        assert torch.equal(marg, _rank_matched(marg, ty_step))
    assert st.graphed
