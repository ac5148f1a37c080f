"""GPU: the product's step driver (frx/ddp.py: DataParallelStep) on the real engine -- hipGraph replay against the
eager step, the data-parallel segment structure on a one-rank RCCL group against the single graph, SphereFace's
annealing lambda through a replayed graph, the optimizer-state round trip with torch.optim.SGD, the uint8 input
path, and the input / label guards of ADVICE r1."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _eng(kind, N, C, dtype, seed=0):
    from frx import engine as E
    return E.FaceEngine(kind, C, N, dtype=dtype, device=DEV, seed=seed)


def _batches(n, N, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [((torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).to(DEV), torch.randint(0, C, (N,), generator=g).to(DEV))
            for _ in range(n)]


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("kind", ["arcface", "sphereface", "elastic_cos"])
def test_graph_replay_equals_the_eager_step(kind):
    """Same init, same batches: 4 steps through DataParallelStep (step 0 eager, graphs captured before step 1, then
    replays) against 4 eager FaceEngine.train_step calls.  fp32 mode: the two differ only by the order of fp32 atomics.
    SphereFace's lambda changes every step (criterion.py:58-60): inside the graph it is read from device state."""
    from frx import ddp, ops
    N, C, lr = 8, 64, 0.01
    a, b = _eng(kind, N, C, ops.F32), _eng(kind, N, C, ops.F32)
    batches = _batches(4, N, C)
    if kind == "sphereface":
        # SphereFace's k = floor(m theta / pi) (criterion.py:88-89) makes the loss a STEP function of the target angle, and a
        # random head puts every target cosine within 0.05 of the k boundary cos = 0 (m = 2): two trajectories that differ by
        # the order of their fp32 atomics then flip k on some row by step 3 (4 of 24 runs 2-4 % apart).
        # Deterministic instead of loose: one batch with labels 0..N-1 whose class rows start ON the batch's own features
        # (target cosine 1, k = 0), checked below to stay far from every k boundary through the four steps.  (Measured with
        # that in place: at lr 0.01 the two trajectories still sat 2.5 % apart at step 3 -- SphereFace's logits are scaled by
        # the feature NORM, not by a fixed s, so the fp32-atomics-order difference of two N = 8 trajectories shows in the loss
        # sooner than with the normalised heads; k flips were not the cause.  A fifth of the step size keeps two correct
        # trajectories inside 2e-2 for four steps; a wrong lambda or a stale graph input is O(1).)
        lr = 0.002
        c = _eng(kind, N, C, ops.F32)
        x0 = batches[0][0]
        c.net.training = True
        f = torch.nn.functional.normalize(c.net.forward(x0).clone(), dim=1)
        y0 = torch.arange(N, device=DEV)
        for e in (a, b):
            w = e.head_w()
            w[:N] = f * w[:N].norm(dim=1, keepdim=True)
        batches = [(x0, y0)] * 4
        del c
    st = ddp.DataParallelStep(a)
    assert st.segments() == [["forward", "upper", "lower", "update"]]
    for i, (x, y) in enumerate(batches):
        oa = st.step(x, y, lr)
        if kind == "elastic_cos":
            b.t.copy_(a.t)                         # the eager twin replays the margins the stepper drew for this step
            b.net.training = True
            b.net.zero_grad()
            ob = b.forward_loss(x, y, sample=False)
            b.backward(y)
            b.net.sgd_step(lr)
        else:
            ob = b.train_step(x, y, lr)
        assert st.graphed == (i >= 1)
        if kind == "sphereface":                   # no target angle near a k boundary (cos = 0 for m = 2; +-0.707, 0 for m = 4)
            for e in (a, b):
                ty = (torch.nn.functional.normalize(e.net.feats, dim=1) * torch.nn.functional.normalize(e.head_w()[:N], dim=1)).sum(1)
                assert ty.min().item() > 0.8, (i, ty.tolist())
        # (two trajectories of an N = 8 fp32 net drift apart chaotically from the order of their fp32 atomics: the
        # comparison is tight through the first REPLAYED step and only a sanity bound afterwards)
        assert oa["loss"].item() == pytest.approx(ob["loss"].item(), rel=1e-4 if i < 2 else 2e-2), (i, kind)
        if i == 1:
            assert _rel(a.net.params, b.net.params) < 1e-3
            assert _rel(a.net.mom, b.net.mom) < 2e-2
    assert torch.isfinite(a.net.params).all()
    if kind == "sphereface":
        assert a.sphere_iter == b.sphere_iter == 4
    # the Python bookkeeping a replay skips is redone (post_replay): eval sees the CURRENT weights and running statistics.
    # (b takes a's state first: the two trajectories differ by ~1e-3, which eval-mode BN after four steps amplifies)
    x = _batches(1, N, C, seed=9)[0][0]
    stale = b.embed(x).clone()                       # b's eval affine is now built for b's own state
    for dst, src in ((b.net.params, a.net.params), (b.net.running_mean, a.net.running_mean), (b.net.running_var, a.net.running_var)):
        dst.copy_(src)
    b.net.sync_weights()
    b.net.stats_version = getattr(b.net, "stats_version", 0) + 1
    fa, fb = a.embed(x).clone(), b.embed(x).clone()
    assert torch.equal(fa, fb), (fa - fb).abs().max().item()
    assert not torch.equal(fb, stale)


@pytest.mark.parametrize("kind,bf16", [("curricular", False), ("arcface", False), ("arcface", True)])
def test_data_parallel_segments_on_a_one_rank_group_equal_the_single_graph(kind, bf16):
    """The multi-GPU structure -- graph segments with RCCL all-reduces (and, for CurricularFace, the target-cosine
    exchange) between them -- rehearsed on ONE GPU in a one-rank process group, against the single-graph step."""
    import torch.distributed as dist
    from frx import ddp, ops
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    N, C, lr = 8, 64, 0.01
    a, b = _eng(kind, N, C, ops.F32), _eng(kind, N, C, ops.F32)
    sa = ddp.DataParallelStep(a, split=True, bf16_buckets=bf16)
    sb = ddp.DataParallelStep(b)
    assert sa.multi and not sb.multi and sa.bf16 == bf16
    # forward [| ty exchange] head | upper | lower | update: the head + fc gradients travel as a bucket of their own
    assert sa.head_bucket and len(sa.segments()) == (5 if kind == "curricular" else 4)
    assert set(a.grad_ranges()) == {"head", "upper", "lower"} and a.grad_ranges()["head"] == [(a.net.fc_w_off, a.net.n_params)]
    for i, (x, y) in enumerate(_batches(3, N, C, seed=2)):
        oa, ob = sa.step(x, y, lr), sb.step(x, y, lr)
        assert oa["loss"].item() == pytest.approx(ob["loss"].item(), rel=(1e-4 if i < 2 else 2e-2) * (20 if bf16 else 1))
        if i == 1:
            assert _rel(a.net.params, b.net.params) < (2e-2 if bf16 else 1e-3)
    assert sa.graphed and sb.graphed
    if kind == "curricular":
        assert a.t.item() == pytest.approx(b.t.item(), rel=1e-2) and a.t.item() != 0.0      # (after the chaotic third step)


def test_optimizer_state_round_trip_with_torch_sgd(tmp_path):
    """Reference checkpoints carry torch.optim.SGD's state (momentum_buffer per parameter, model_utils.py:58-65,126-132).
    (1) FusedSGD.state_dict() is in that format: a plain torch.optim.SGD over the same parameter list loads it and its
    buffers equal the engine's momentum; (2) a state_dict written by torch.optim.SGD loads into FusedSGD and lands in the
    engine's flat momentum buffer (NCHW -> KRSC, stem padding slots zero); (3) one further step from the same state
    moves the parameters identically."""
    from test_gpu_dropin import _batch, _mk
    from utils import model_utils as MU
    C = 24
    m = _mk("ArcFaceNet", C, "f32", seed=11)
    crit = nn.CrossEntropyLoss().to(DEV)
    args = types.SimpleNamespace(lambda_g=0.0, print_freq=10)
    data = [tuple(t.cpu() for t in _batch(6, C, 20 + i)) for i in range(3)]
    opt = MU.make_optimizer(m, 0.02)
    MU.train_model(m, data, crit, opt, MU.GradScaler(enabled=False), DEV, 1, 1, args)
    sd = opt.state_dict()
    params = list(m.parameters())
    assert set(sd["state"]) == set(range(len(params)))
    assert all(tuple(sd["state"][i]["momentum_buffer"].shape) == tuple(p.shape) for i, p in enumerate(params))
    assert "frx_momentum" not in sd
    torch.save({"optimizer_state_dict": sd}, tmp_path / "o.pth")
    sd = torch.load(tmp_path / "o.pth", weights_only=True)["optimizer_state_dict"]       # what load_latest_checkpoint does
    # (1) torch's own SGD accepts it
    tsgd = torch.optim.SGD(params, lr=0.02, momentum=0.9, weight_decay=5e-4)
    tsgd.load_state_dict(sd)
    net = m._primary.net
    conv = next(c for c in net.convs if c.name == "layer2.0.conv2")
    idx = [i for i, (n_, _) in enumerate(m.named_parameters()) if n_ == "backbone.layer2.0.conv2.weight"][0]
    buf = tsgd.state[params[idx]]["momentum_buffer"]
    assert torch.equal(buf, net.w_grad(conv, net.mom).permute(0, 3, 1, 2)) and buf.abs().max().item() > 0
    # (2) a second model resumes from torch's state: its engine momentum equals the first engine's
    m2 = _mk("ArcFaceNet", C, "f32", seed=12)
    m2.load_state_dict(m.state_dict())
    opt2 = MU.make_optimizer(m2, 0.5)
    opt2.load_state_dict(tsgd.state_dict())             # written by torch.optim.SGD
    assert opt2.param_groups[0]["lr"] == 0.02
    x, y = _batch(6, C, 40)
    # (3) next step: fused on m2 vs torch SGD on m (autograd-compatible path)
    MU.train_model(m2, [(x.cpu(), y.cpu())], crit, opt2, MU.GradScaler(enabled=False), DEV, 2, 2, args)
    MU.train_model(m, [(x.cpu(), y.cpu())], crit, tsgd, MU.GradScaler(enabled=False), DEV, 2, 2, args)
    assert _rel(m2._primary.net.params, m._primary.net.params) < 1e-5
    stem_m = net.w_grad(net.stem, m2._primary.net.mom)
    assert stem_m[:, :, 7, :].abs().max().item() == 0 and stem_m[..., 3].abs().max().item() == 0


def test_uint8_batches_step_exactly_like_the_fp32_transform(monkeypatch):
    """f2: datasets may hand uint8 HWC batches (dataset.uint8_hwc); ToTensor + Normalize then run inside
    frx_input_prep.  The stem input, the embeddings and the loss are BIT-identical to feeding the reference's
    fp32 transform of the same pixels (model_utils.py:539-547); train_model takes both through the same loop.
    (Bit-identity of two training runs needs the bit-reproducible BatchNorm sums: FRX_BN_DETERMINISTIC=1.)"""
    monkeypatch.setenv("FRX_BN_DETERMINISTIC", "1")
    from test_gpu_dropin import _mk
    from utils import model_utils as MU
    from utils.dataset import default_transform, uint8_hwc
    C, N = 16, 8
    rng = np.random.RandomState(3)
    raw = rng.randint(0, 256, (N, 112, 112, 3), dtype=np.uint8)
    u8 = torch.stack([uint8_hwc(r) for r in raw])                      # [N,112,112,3] uint8
    f32 = torch.stack([default_transform(r) for r in raw])             # [N,3,112,112] fp32 in [-1,1]
    y = torch.from_numpy(rng.randint(0, C, N)).long()
    ma, mb = _mk("ArcFaceNet", C, "bf16", seed=21), _mk("ArcFaceNet", C, "bf16", seed=22)
    mb.load_state_dict(ma.state_dict())
    crit = nn.CrossEntropyLoss().to(DEV)
    args = types.SimpleNamespace(lambda_g=0.0, print_freq=1)
    la = MU.train_model(ma, [(u8, y)], crit, MU.make_optimizer(ma, 0.01), MU.GradScaler(enabled=False), DEV, 1, 1, args)
    lb = MU.train_model(mb, [(f32, y)], crit, MU.make_optimizer(mb, 0.01), MU.GradScaler(enabled=False), DEV, 1, 1, args)
    assert torch.equal(ma._primary.net.xin, mb._primary.net.xin), "uint8 and fp32 staging must give the same stem input"
    assert torch.equal(ma._primary.net.feats, mb._primary.net.feats)
    assert la == lb and np.isfinite(la)
    # eval takes both input forms too.  (The two models took one update each, and weight gradients are summed with fp32
    # atomics in no fixed order: give them the same weights again before asking for bit-equal embeddings.)
    mb.load_state_dict(ma.state_dict())
    ma.eval(); mb.eval()
    assert torch.equal(ma(u8.to(DEV)), mb(f32.to(DEV)))


def test_inputs_of_another_size_and_labels_out_of_range_are_caught():
    """a 224x224 batch must raise instead of overrunning the 112x112 staging buffer (engine AND C ABI);
    a label outside [0, C) must not become an out-of-bounds access -- the loss turns NaN, parameters stay intact
    up to that NaN step being visible at the next loss read."""
    from frx import ops
    from frx._lib import FrxError
    N, C = 4, 10
    eng = _eng("arcface", N, C, ops.BF16)
    with pytest.raises(FrxError, match="112x112"):
        eng.net.forward(torch.zeros(N, 3, 224, 224, device=DEV))
    with pytest.raises(FrxError, match="112x112"):
        eng.net.forward(torch.zeros(N, 250, 250, 3, dtype=torch.uint8, device=DEV))
    with pytest.raises(FrxError):
        ops.input_prep(ops.BF16, torch.zeros(N, 3, 224, 224, device=DEV), eng.net.xin)        # the C ABI's own check
    with pytest.raises(FrxError, match="3 colour"):
        eng.net.forward(torch.zeros(N, 1, 112, 112, device=DEV))
    x = _batches(1, N, C)[0][0]
    good = torch.tensor([0, 3, 9, 1], device=DEV)
    assert torch.isfinite(eng.forward_loss(x, good)["loss"]).all()
    for bad in (torch.tensor([0, 3, 10, 1], device=DEV), torch.tensor([-1, 3, 9, 1], device=DEV)):
        out = eng.forward_loss(x, bad)
        eng.backward(bad)
        torch.cuda.synchronize()
        assert torch.isnan(out["loss"]).all()
    for kind in ("curricular", "vpl_arcface"):
        e2 = _eng(kind, N, C, ops.BF16)
        before = e2.t.clone()
        out = e2.forward_loss(x, torch.tensor([0, 3, 12345678, 1], device=DEV))
        torch.cuda.synchronize()
        assert torch.isnan(out["loss"]).all()
        if kind == "vpl_arcface":                 # the class memory is only written for the valid labels
            assert torch.isfinite(e2.t).all() and (e2.t[:C * 512].view(C, 512)[[0, 3, 1]].abs().sum(1) > 0).all()
