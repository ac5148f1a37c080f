"""GPU: the class-sharded head's HIP phases (frx_head_shard_cos / _rows / _rescale / _finish + frx_head_bwd in shard mode)
against oracle/heads.py (float64, the unsharded closed form) on the same inputs: the shards are run one after another
on ONE GPU and the collectives of frx/ddp.py: sharded_plan are done by hand (sum / max over the shards' buffers), which
is the same arithmetic.  Loss / lse within the north-star 1e-3, gradients within 1e-3 of their scale, also at 128 x 85 000.
Then the engine wiring (FaceEngine(shard=...) through DataParallelStep on a one-rank RCCL group) against the plain engine.
Reference for the partition: criterion.py:268-278 (the dormant device_id chunking)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
D = 512


def _inputs(kind, N, C, seed=0):
    from frx import ops
    g = torch.Generator().manual_seed(seed)
    cd = ops.__dict__[kind] in ops.W_CD_KINDS
    w = torch.randn(C, D, generator=g) if cd else torch.randn(D, C, generator=g)
    y = torch.randint(0, C, (N,), generator=g)
    x = torch.randn(N, D, generator=g)
    wc = torch.nn.functional.normalize(w if cd else w.t(), dim=1)
    for i in range(0, N, 3):                       # a third of the rows aligned with their class: both margin branches fire
        x[i] = wc[y[i]] * 4 + 0.3 * torch.randn(D, generator=g)
    return x.to(DEV), w.to(DEV), y.to(DEV), cd


HYPER = {"ARC": (64.0, 0.5), "COS": (64.0, 0.35), "CURR": (64.0, 0.5), "SPHERE": (1.0, 2.0), "MV_ARC": (32.0, 0.35)}


def _oracle(kind, x, w, y, t0, sphere_iter=6):
    """float64 closed form of the UNSHARDED head (oracle/heads.py; reference criterion.py:263-301 and siblings)"""
    from oracle import heads as H
    K = getattr(H, kind)
    hy = H.HeadHyper.default(K)
    st = H.HeadState(iter=sphere_iter, t=t0)
    ref = H.head_forward_backward(K, x.cpu().numpy(), w.cpu().numpy(), y.cpu().numpy(), hy, st, dtype=np.float64)
    return ref, st


def _run_shards(kind, world, x, w, y, cd, t0, lamb):
    """the shard phases of `world` shards one after another on ONE GPU; the collectives of frx/ddp.py: sharded_plan done
    by hand (sum / max over the shards' buffers: the same arithmetic).  Returns (per-shard outputs, dx, [(c0, cl, dw)], t)."""
    from frx import ops
    K = ops.__dict__[kind]
    N, C = x.shape[0], (w.shape[0] if cd else w.shape[1])
    s_, m_ = HYPER[kind]
    p = (1.12,) if kind == "MV_ARC" else ()
    Cs = -(-C // world)
    shards = []
    for r in range(world):
        c0, cl = r * Cs, min(C, (r + 1) * Cs) - r * Cs
        wl = (w[c0:c0 + cl] if cd else w[:, c0:c0 + cl]).contiguous()
        c = ops.HeadContext(K, N, D, cl, s_, m_, 0.01, device=DEV, p=p, class_offset=c0)
        if K == ops.SPHERE:
            c.desc.lamb = lamb
        shards.append(dict(ctx=c, w=wl, c0=c0, cl=cl, t=torch.full((1,), t0, device=DEV), ty=torch.zeros(N, device=DEV),
                           part=torch.zeros(3, N, device=DEV)))
    for sh in shards:
        ops.head_shard_cos(sh["ctx"], x, sh["w"], y, sh["ty"])
    ty_g = sum(sh["ty"] for sh in shards)                                   # all-reduce SUM
    owners = sum((sh["ty"] != 0).int() for sh in shards)
    assert int(owners.max()) <= 1, "two shards claimed the same row's target"
    for sh in shards:
        ops.head_shard_rows(sh["ctx"], y, ty_g, sh["part"], state_t=sh["t"])
    gmax = torch.stack([sh["part"][0] for sh in shards]).max(0).values.contiguous()      # all-reduce MAX
    for sh in shards:
        ops.head_shard_rescale(sh["part"][0], gmax, sh["part"][1])
    gsum = sum(sh["part"][1] for sh in shards).contiguous()                 # all-reduce SUM
    grank = sum(sh["part"][2] for sh in shards).contiguous()
    outs, dws = [], []
    dx = torch.zeros_like(x)
    for sh in shards:
        outs.append(ops.head_shard_finish(sh["ctx"], gmax, gsum, grank, state_t=sh["t"]))
        pdx, pdw = ops.head_backward(sh["ctx"], x, sh["w"], y, state_t=sh["t"])
        dx += pdx                                                           # reduce-scatter SUM (all rows here)
        dws.append((sh["c0"], sh["cl"], pdw))
    return outs, dx, dws, [sh["t"].item() for sh in shards]


def _check_vs_oracle(kind, outs, dx, dws, ts, ref, st, cd, t0):
    for o in outs:                                  # every rank ends with the same loss / lse / top-k
        assert abs(o["loss"].item() - ref.loss) < 1e-3, (o["loss"].item(), ref.loss)
        np.testing.assert_allclose(o["lse"].cpu().numpy(), ref.lse, atol=1e-3)
        assert tuple(o["topk"].tolist()) == (ref.top1, ref.top5)
        np.testing.assert_allclose(o["norms"].cpu().numpy(), ref.norms.reshape(-1), rtol=1e-5)
    if kind == "CURR":
        for t in ts:
            assert t == pytest.approx(st.t, abs=1e-6) and t != t0
    sdx, sdw = np.abs(ref.dx).max(), np.abs(ref.dw).max()
    np.testing.assert_allclose(dx.cpu().numpy(), ref.dx, atol=1e-3 * sdx, rtol=0)
    for c0, cl, pdw in dws:
        want = ref.dw[c0:c0 + cl] if cd else ref.dw[:, c0:c0 + cl]
        np.testing.assert_allclose(pdw.cpu().numpy(), want, atol=1e-3 * sdw, rtol=0)


@pytest.mark.parametrize("kind,world", [("ARC", 2), ("COS", 3), ("CURR", 2), ("SPHERE", 2), ("MV_ARC", 2)])
def test_shard_phases_vs_float64_oracle(kind, world):
    """sharded HIP head (N = 48, C = 100 over 2-3 shards, one of them ragged) against the float64 oracle of the unsharded
    head: loss / lse within the north-star 1e-3, top-k equal, dX and every shard's dW columns within 1e-3 of their scale,
    CurricularFace's EMA identical on every shard."""
    x, w, y, cd = _inputs(kind, 48, 100)
    t0 = 0.05
    ref, st = _oracle(kind, x, w, y, t0)
    outs, dx, dws, ts = _run_shards(kind, world, x, w, y, cd, t0, st.lamb)
    _check_vs_oracle(kind, outs, dx, dws, ts, ref, st, cd, t0)


def test_sharded_curricular_128x85000_vs_float64_oracle():
    """configs[3]'s head at its per-GPU size, class-sharded 8 ways (85 000 = 8 x 10 625), against the float64 oracle"""
    g = torch.Generator().manual_seed(7)
    N, C = 128, 85000
    w = torch.randn(D, C, generator=g) * 0.01
    y = torch.randint(0, C, (N,), generator=g)
    x = torch.randn(N, D, generator=g)
    wc = torch.nn.functional.normalize(w.t()[y], dim=1)
    x[::3] = wc[::3] * 4 + 0.3 * torch.randn(len(x[::3]), D, generator=g)
    x, w, y = x.to(DEV), w.to(DEV), y.to(DEV)
    ref, st = _oracle("CURR", x, w, y, 0.05)
    outs, dx, dws, ts = _run_shards("CURR", 8, x, w, y, False, 0.05, 0.0)
    _check_vs_oracle("CURR", outs, dx, dws, ts, ref, st, False, 0.05)


def test_sharded_mode_rejects_heads_with_batch_wide_state():
    from frx import ops
    from frx._lib import FrxError
    with pytest.raises(FrxError, match="class-sharded"):
        ops.HeadContext(ops.ADA, 8, D, 10, 64.0, 0.4, device=DEV, p=(0.333, 0.99), class_offset=0)


@pytest.mark.parametrize("kind", ["curricular", "arcface"])
def test_engine_with_a_sharded_head_on_a_one_rank_group_equals_the_plain_engine(kind):
    """FaceEngine(shard=(0, 1)) through the shipped step driver (eight plan stages, collectives degenerate to copies) against
    the replicated-head engine: same loss, same parameters after the first replayed step."""
    import torch.distributed as dist
    from frx import ddp, engine as E, ops
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    N, C = 8, 96
    a = E.FaceEngine(kind, C, N, dtype=ops.F32, device=DEV, seed=0, shard=(0, 1))
    b = E.FaceEngine(kind, C, N, dtype=ops.F32, device=DEV, seed=0)
    assert torch.equal(a.head_w(), b.head_w()) and torch.equal(a.net.params[:a.net.extra_off], b.net.params[:b.net.extra_off])
    sa, sb = ddp.DataParallelStep(a), ddp.DataParallelStep(b)
    assert len(sa.segments()) == 6 and sa.segments()[0] == ["backbone"]          # five exchange points + [upper, lower, update]
    rng = a.grad_ranges()
    assert max(hi for v in rng.values() for _, hi in v) == a.net.extra_off, "the head columns must stay off the wire"
    g = torch.Generator().manual_seed(4)
    for i in range(3):
        x = (torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).to(DEV)
        y = torch.randint(0, C, (N,), generator=g).to(DEV)
        oa, ob = sa.step(x, y, 0.01), sb.step(x, y, 0.01)
        assert oa["loss"].item() == pytest.approx(ob["loss"].item(), rel=1e-4 if i < 2 else 2e-2)
        assert oa["topk"].tolist() == ob["topk"].tolist() or i == 2
        if i == 1:
            rel = ((a.net.params - b.net.params).norm() / b.net.params.norm()).item()
            assert rel < 1e-3, rel
    assert sa.graphed and sb.graphed
    if kind == "curricular":
        assert a.t.item() == pytest.approx(b.t.item(), rel=1e-2) and a.t.item() != 0      # (after the chaotic third step)
    full = a.gather_head_weight()
    assert torch.equal(full, a.head_w())
