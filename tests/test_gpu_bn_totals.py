"""GPU: BatchNorm statistics as replicated totals (csrc/bn_tot.h: producers add per-channel sums with float atomics into R
rows, consumers derive scale / shift and the backward coefficients themselves, one batched closing launch per pass) against
the bit-reproducible form (partial rows + one finalize launch per layer) of the same engine -- which the whole-net tests
hold to the CPU oracle.  Everything the two forms hand to the rest of the step must agree to the rounding of the sums."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(n, c, kind="arcface"):
    from frx import engine as E, ops
    engs = []
    for det in ("1", "0"):
        os.environ["FRX_BN_DETERMINISTIC"] = det
        try:
            engs.append(E.FaceEngine(kind, c, n, dtype=ops.BF16, device=DEV, seed=0))
        finally:
            os.environ.pop("FRX_BN_DETERMINISTIC", None)
    det, fus = engs
    assert not det.net.fused_bn and fus.net.fused_bn
    assert torch.equal(det.net.params, fus.net.params)
    return det, fus


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("n", [16, 256])
def test_forward_is_self_consistent_and_tracks_the_deterministic_form(n):
    """Two bf16 trajectories that differ in the last bit of a scale diverge through 50 layers of rounding flips, so an
    element-wise comparison of two ENGINES measures chaos, not correctness (measured: statistics 1e-7 apart at the stem,
    1-5 % at layer4).  The forward is therefore checked inside ONE engine: every layer's mean / invstd (as written by the
    batched closing launch from the totals) against torch reductions of that layer's own raw output, every block output
    against the merge recomputed from the engine's own constants; the deterministic engine only has to agree in the
    aggregate (loss, running statistics)."""
    det, fus = _pair(n, 1000)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(n, 3, 112, 112, generator=g) * 2 - 1).to(DEV)
    y = torch.randint(0, 1000, (n,), generator=g).to(DEV)
    outs = []
    for e in (det, fus):
        e.net.training = True
        outs.append(e.forward_loss(x, y))
    nd, nf = det.net, fus.net
    s = nd.stem                       # identical inputs at the stem: agreement to the rounding of a float sum
    for name in ("bn_mean", "bn_invstd", "bn_scale", "bn_shift"):
        a, b = nf._bn(getattr(nf, name), s), nd._bn(getattr(nd, name), s)
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-7, name
    for c in nf.convs:
        yy = c.y.float().reshape(-1, c.Co)
        m, v = yy.mean(0), yy.var(0, unbiased=False)
        inv = (v + 1e-5).rsqrt()
        assert (nf._bn(nf.bn_mean, c) - m).abs().max().item() <= 1e-5 * m.abs().max().item() + 1e-6, c.name
        assert (nf._bn(nf.bn_invstd, c) - inv).abs().max().item() <= 1e-4 * inv.abs().max().item(), c.name
        sc = nf.gamma(c) * nf._bn(nf.bn_invstd, c)
        assert torch.equal(nf._bn(nf.bn_scale, c), sc), c.name
    xin = nf.pool_out
    for b in nf.blocks:
        s3, h3 = nf._bn(nf.bn_scale, b.conv3), nf._bn(nf.bn_shift, b.conv3)
        idn = xin.float() if b.down is None else b.down.y.float() * nf._bn(nf.bn_scale, b.down) + nf._bn(nf.bn_shift, b.down)
        ref = torch.relu(b.conv3.y.float() * s3 + h3 + idn)
        assert (b.out.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item(), b.conv3.name      # (bf16 output)
        xin = b.out
    assert _rel(nf.running_mean, nd.running_mean) < 2e-2 and _rel(nf.running_var, nd.running_var) < 2e-2
    assert abs(outs[1]["loss"].item() - outs[0]["loss"].item()) < 2e-2 * abs(outs[0]["loss"].item())
    # the totals are zeroed by the closing launch: a second forward gives the same statistics again (no accumulation)
    assert float(nf.bn_tot_f.abs().max()) == 0.0
    m1 = nf.bn_mean.clone()
    fus.forward_loss(x, y)
    assert (nf._bn(nf.bn_mean, s) - nf._bn(m1, s)).abs().max().item() <= 2e-5 * m1.abs().max().item() + 1e-6
    assert float(nf.bn_tot_f.abs().max()) == 0.0


@pytest.mark.parametrize("n", [16, 256])
def test_backward_with_totals_equals_the_deterministic_backward_of_the_same_forward(n):
    """ONE engine, one forward; the backward runs twice from that state -- replicated totals, then (flag flipped) partial
    rows + finalize launches.  Same activations, same masks, same upstream gradient: every weight / gamma / beta gradient
    must agree to the rounding of the BatchNorm sums (and the few bf16 flips of dz / dy that follow from it)."""
    _, fus = _pair(n, 1000)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, 3, 112, 112, generator=g) * 2 - 1).to(DEV)
    y = torch.randint(0, 1000, (n,), generator=g).to(DEV)
    df = (torch.randn(n, 512, generator=g) * 1e-3).to(DEV)
    net = fus.net
    net.training = True
    fus.forward_loss(x, y)
    net.zero_grad(); net.backward(df)
    g_tot = net.grads.clone()
    coefs_tot = [[c.clone() if c is not None else None for c in b.coefs] for b in net.blocks]
    assert float(net.bn_tot_b.abs().max()) == 0.0 and float(net.bn_tot_f.abs().max()) == 0.0
    #  a check of the DEFAULT backward that compares no two runs: dbeta = sum dz, dgamma = sum dz * xhat of every
    # bn3 and every projection BatchNorm -- the layers whose dz outlives the step (the block's masked output gradient) --
    # against torch reductions of the engine's OWN dz / y buffers.  A stale or double-counted total, a totals row that was
    # not zeroed, or a missed fork / join of the projection branch shows up here as O(1), rounding as 1e-5.
    for b in net.blocks:
        for c in (b.conv3, b.down):
            if c is None:
                continue
            dz, yy = b.dz3.float().reshape(-1, c.Co), c.y.float().reshape(-1, c.Co)
            xhat = (yy - net._bn(net.bn_mean, c)) * net._bn(net.bn_invstd, c)
            for got, terms in ((net.beta(c, g_tot), dz), (net.gamma(c, g_tot), dz * xhat)):
                tol = 2e-4 * terms.abs().sum(0) + 1e-7
                assert ((got - terms.sum(0)).abs() <= tol).all(), (c.name, float(((got - terms.sum(0)).abs() / tol).max()))
    net.fused_bn = False
    try:
        net.zero_grad(); net.backward(df)
    finally:
        net.fused_bn = True
    g_det = net.grads
    assert torch.isfinite(g_tot).all() and torch.isfinite(g_det).all()
    worst = 0.0
    for c in net.convs:
        r = _rel(net.w_grad(c, g_tot), net.w_grad(c, g_det))
        worst = max(worst, r)
        # (measured: 1e-4 at layer4, growing towards the stem as last-bit differences of alpha / beta / gam flip bf16
        # roundings of dz / dy along 50 layers of backward: 1-2 % in dW, up to 7 % in the stem's 64-element dbeta)
        assert r < 5e-2, (c.name, "dW", r)
        assert _rel(net.gamma(c, g_tot), net.gamma(c, g_det)) < 0.15, (c.name, "dgamma")
        assert _rel(net.beta(c, g_tot), net.beta(c, g_det)) < 0.15, (c.name, "dbeta")
        assert float(net.w_grad(c, g_tot).abs().max()) > 0
    for bt, b in zip(coefs_tot, net.blocks):
        for ct, cd in zip(bt, b.coefs):
            if cd is not None:
                assert _rel(ct, cd) < 2e-2
    print(f"batch {n}: worst relative dW difference between the two backward forms {worst:.3e}")


def test_projection_branch_on_a_side_stream_equals_the_in_line_order():
    """ the projection branch of a layer's first block forks onto a side stream inside the step (default) or runs in
    line (branch_stream = None).  ONE engine, one forward; the backward runs twice from that state with the same upstream
    gradient, once per order: same activations, same masks -- the gradients may differ only by the arrival order of float
    atomics and the bf16 flips that follow (the criterion of the totals-vs-deterministic test above).  A missed join would
    leave a projection's gradients, or the addend conv1's input gradient takes from it, stale or half-written: O(1).
    (Two ENGINES cannot be compared this way: their forwards differ by atomics order, which this random-init network
    amplifies to ~10 % in the embeddings -- and the head's softmax at scale 64 turns that into an unrelated gradient.)
    The forward fork / join is covered by test_forward_is_self_consistent (every block output against its own operands)."""
    _, fus = _pair(16, 1000)
    net = fus.net
    assert net.branch_stream is not None
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(16, 3, 112, 112, generator=g) * 2 - 1).to(DEV)
    y = torch.randint(0, 1000, (16,), generator=g).to(DEV)
    df = (torch.randn(16, 512, generator=g) * 1e-3).to(DEV)
    net.training = True
    fus.forward_loss(x, y)
    net.zero_grad(); net.backward(df)
    torch.cuda.synchronize()
    g_side = net.grads.clone()
    side, net.branch_stream = net.branch_stream, None
    try:
        net.zero_grad(); net.backward(df)
        torch.cuda.synchronize()
    finally:
        net.branch_stream = side
    worst = 0.0
    for c in net.convs:
        r = _rel(net.w_grad(c, g_side), net.w_grad(c))
        worst = max(worst, r)
        assert r < 5e-2, (c.name, r)
        assert float(net.w_grad(c, g_side).abs().max()) > 0
    print(f"worst relative dW difference, projection branch on a side stream vs in line: {worst:.3e}")


def test_step_driver_with_totals_trains():
    from frx import ddp
    det, fus = _pair(32, 200)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(32, 3, 112, 112, generator=g) * 2 - 1).to(DEV)
    y = torch.randint(0, 200, (32,), generator=g).to(DEV)
    sd, sf = ddp.DataParallelStep(det), ddp.DataParallelStep(fus)
    ld = [sd.step(x, y, 0.002)["loss"].item() for _ in range(16)]
    lf = [sf.step(x, y, 0.002)["loss"].item() for _ in range(16)]
    assert sf.graphed and all(np.isfinite(lf))
    for i in range(4):                     # the two forms follow each other while the trajectories are still close
        assert lf[i] == pytest.approx(ld[i], rel=2e-2), (i, ld, lf)
    assert lf[-1] < lf[0] - 0.5 and ld[-1] < ld[0] - 0.5, (ld, lf)

# ------------------------------------------------------------------------------------------------------------------
# every entry point of the replicated-totals form against its partial-rows twin on IDENTICAL inputs: a partial buffer with
# exactly R rows is at the same time a valid totals buffer, so the two forms see the same sums and must agree bit for bit
# (consumers) or to float-sum rounding (producers: atomics vs stores)
# ------------------------------------------------------------------------------------------------------------------
def _rand(dtype, *shape, seed=0, scale=1.0):
    from frx import ops
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(ops.TORCH_DT[dtype]).to(DEV)


def _fwd_consts(part, count, gamma, beta):
    """frx_bn_finalize on an [R][2][C] buffer -> (mean, invstd, scale, shift)"""
    from frx import ops
    R, _, Cc = part.shape
    outs = [torch.zeros(Cc, device=DEV) for _ in range(4)]
    ops.bn_finalize(part, R, Cc, count, gamma, beta, None, None, *outs)
    return outs


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(64, 64, 3, 1, 28), (64, 256, 1, 1, 28), (256, 256, 3, 1, 7), (1024, 256, 1, 1, 7), (512, 512, 3, 2, 7)],
                         ids=lambda s: "x".join(map(str, s)))
def test_conv_fwd_tot_equals_conv_fwd(dtype, shape, monkeypatch):
    from frx import ops
    Ci, Co, k, stride, Hi = shape
    N, R = 24, 8
    # (bit equality of the two forms needs one main loop under both: at this batch the partial-rows form of the 3x3
    # layers takes 64-pixel row tiles, the totals form the 128-pixel patch tile, whose fp32 sums run in another order --
    # the patch kernel has its own comparisons: test_gpu_conv.py::test_patch_3x3_*)
    monkeypatch.setenv("FRX_CONV3X3", "0")
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, k, k, stride, k // 2)
    x, w = _rand(dtype, N, Hi, Hi, Ci, seed=1), _rand(dtype, Co, k, k, Ci, seed=2, scale=(Ci * k * k) ** -0.5)
    g = torch.Generator().manual_seed(3)
    gamma, beta = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    gamma[::3] *= -1
    count = 5000
    tin = torch.zeros(R, 2, Ci, device=DEV)                     # statistics of the INPUT's BatchNorm, as R rows
    tin[:, 0] = (torch.randn(R, Ci, generator=g) * 30).to(DEV)
    tin[:, 1] = (torch.rand(R, Ci, generator=g) * 800 + 400).to(DEV)
    mean, invstd, sc, sh = _fwd_consts(tin, count, gamma, beta)
    # reference form
    rows = ops.conv_stat_rows(d)
    y0 = torch.empty(N, d.Ho, d.Wo, Co, dtype=x.dtype, device=DEV)
    part = torch.zeros(rows, 2, Co, device=DEV)
    ops.conv_fwd(d, x, w, y0, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)
    # totals form: constants derived in the prologue, statistics added into R rows
    y1 = torch.empty_like(y0)
    tout = torch.zeros(R, 2, Co, device=DEV)
    ops.conv_fwd_tot(d, x, w, y1, in_bn=ops.bn_tot(tin, R, count, gamma, beta=beta), stat_totals=tout, stat_replicas=R)
    assert torch.equal(y1, y0), "derived scale / shift must be the finalize kernel's, bit for bit"
    a, b = tout.sum(0), part.sum(0)
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    used = (tout.abs().sum((1, 2)) > 0).sum().item()
    assert used == min(R, rows), "row tiles must spread over the replicas"
    # a second launch ADDS
    ops.conv_fwd_tot(d, x, w, y1, in_bn=None, stat_totals=tout, stat_replicas=R)
    y2 = torch.empty_like(y0); part2 = torch.zeros_like(part)
    ops.conv_fwd(d, x, w, y2, stat_partial=part2)
    assert torch.equal(y1, y2)
    assert (tout.sum(0) - (b + part2.sum(0))).abs().max().item() <= 2e-5 * (b + part2.sum(0)).abs().max().item()


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("down", [False, True], ids=["identity", "projection"])
@pytest.mark.parametrize("form", ["arrays", "totals"])
@pytest.mark.parametrize("shape", [(256, 64, 28, 8), (256, 128, 28, 5), (512, 128, 14, 16), (1024, 256, 7, 32), (2048, 512, 4, 20), (128, 64, 5, 3), (64, 192, 9, 2)],
                         ids=lambda s: "x".join(map(str, s)))
def test_conv_fwd_merge_equals_merge_then_conv(dtype, down, form, shape):
    """frx_conv_fwd_merge (the residual merge of the block before as the prologue of the next conv1; round 4) against the
    two launches it replaces on the same inputs: block output and mask bits as frx_block_merge_fwd(_mask / _tot) writes them,
    BIT FOR BIT; the conv output bit for bit (same tile, same K order, the staged operand is the same bf16 / fp32 values);
    the statistics to float-sum rounding.  ResNet-50's conv1 shapes of the four layers plus ragged ones (75 and 162 pixel
    rows: tile tails; 192 output columns: the 64-column tile)."""
    from frx import ops
    Ci, Co, H, N = shape
    R, count = 8, N * H * H
    rows = N * H * H
    g = torch.Generator().manual_seed(Ci + Co + H)

    def tot(C_):
        t = torch.zeros(R, 2, C_, device=DEV)
        t[:, 0] = (torch.randn(R, C_, generator=g) * 10).to(DEV); t[:, 1] = (torch.rand(R, C_, generator=g) * 300 + 100).to(DEV)
        ga, be = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.3).to(DEV)
        ga[::5] *= -1
        return t, ga, be
    t3, g3, b3 = tot(Ci); td, gd, bd = tot(Ci)
    _, _, s3, h3 = _fwd_consts(t3, count, g3, b3)
    _, _, sd, hd = _fwd_consts(td, count, gd, bd)
    y3, idn = _rand(dtype, N, H, H, Ci, seed=1), _rand(dtype, N, H, H, Ci, seed=2)
    w = _rand(dtype, Co, 1, 1, Ci, seed=3, scale=Ci ** -0.5)
    V = 8 if dtype == 1 else 4
    d = ops.conv_desc(dtype, N, H, H, Ci, Co, 1, 1, 1, 0)
    # the two launches
    o0 = torch.empty_like(y3); m0 = torch.zeros(rows * Ci // V, dtype=torch.uint8, device=DEV)
    ops.block_merge_fwd(dtype, rows, Ci, y3, s3, h3, idn, o0, sd=sd if down else None, bd=hd if down else None, mask=m0)
    y0 = torch.empty(N, H, H, Co, dtype=y3.dtype, device=DEV)
    tout0 = torch.zeros(R, 2, Co, device=DEV)
    ops.conv_fwd_tot(d, o0, w, y0, in_bn=None, stat_totals=tout0, stat_replicas=R)
    # the fused launch
    o1 = torch.full_like(y3, 7.0); m1 = torch.full_like(m0, 255)
    y1 = torch.empty_like(y0); tout1 = torch.zeros_like(tout0)
    if form == "arrays":
        kw = dict(s3=s3, b3=h3, sd=sd if down else None, bd=hd if down else None)
    else:
        kw = dict(bn3=ops.bn_tot(t3, R, count, g3, beta=b3), bnd=ops.bn_tot(td, R, count, gd, beta=bd) if down else None)
    ops.conv_fwd_merge(d, y3, idn, w, y1, o1, mask=m1, stat_totals=tout1, stat_replicas=R, **kw)
    assert torch.equal(o1, o0), "block output: the merge pass's bits"
    assert torch.equal(m1, m0), "mask bits"
    assert torch.equal(y1, y0), "conv output"
    a, b = tout1.sum(0), tout0.sum(0)
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-6
    # plain epilogue (no statistics), no mask
    y2 = torch.empty_like(y0); o2 = torch.empty_like(y3)
    ops.conv_fwd_merge(d, y3, idn, w, y2, o2, **kw)
    assert torch.equal(y2, y0) and torch.equal(o2, o0)


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("down", [False, True], ids=["identity", "projection"])
def test_merge_and_pool_tot_equal_their_twins(dtype, down):
    from frx import ops
    rows, Cc, R, count = 37 * 49, 256, 8, 1813
    g = torch.Generator().manual_seed(11)

    def tot(C_):
        t = torch.zeros(R, 2, C_, device=DEV)
        t[:, 0] = (torch.randn(R, C_, generator=g) * 10).to(DEV); t[:, 1] = (torch.rand(R, C_, generator=g) * 300 + 100).to(DEV)
        ga, be = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.3).to(DEV)
        return t, ga, be
    t3, g3, b3 = tot(Cc); td, gd, bd = tot(Cc)
    _, _, s3, h3 = _fwd_consts(t3, count, g3, b3)
    _, _, sd, hd = _fwd_consts(td, count, gd, bd)
    y3, idn = _rand(dtype, rows, Cc, seed=1), _rand(dtype, rows, Cc, seed=2)
    V = 8 if dtype == 1 else 4
    o0, o1 = torch.empty_like(y3), torch.empty_like(y3)
    m0 = torch.zeros(rows * Cc // V, dtype=torch.uint8, device=DEV); m1 = torch.zeros_like(m0)
    ops.block_merge_fwd(dtype, rows, Cc, y3, s3, h3, idn, o0, sd=sd if down else None, bd=hd if down else None, mask=m0)
    ops.block_merge_fwd_tot(dtype, rows, Cc, y3, ops.bn_tot(t3, R, count, g3, beta=b3), idn, o1,
                            bnd=ops.bn_tot(td, R, count, gd, beta=bd) if down else None, mask=m1)
    assert torch.equal(o1, o0) and torch.equal(m1, m0)
    if not down:                            # the stem's pool, R = 32
        N, H, C_ = 5, 56, 64
        R2 = 32
        t = torch.zeros(R2, 2, C_, device=DEV)
        t[:, 0] = (torch.randn(R2, C_, generator=g) * 10).to(DEV); t[:, 1] = (torch.rand(R2, C_, generator=g) * 300 + 100).to(DEV)
        ga, be = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.3).to(DEV)
        _, _, sc, sh = _fwd_consts(t, 4000, ga, be)
        yy = _rand(dtype, N, H, H, C_, seed=5)
        Ho = (H + 2 - 3) // 2 + 1
        p0, p1 = torch.empty(N, Ho, Ho, C_, dtype=yy.dtype, device=DEV), torch.empty(N, Ho, Ho, C_, dtype=yy.dtype, device=DEV)
        a0, a1 = torch.zeros(N, Ho, Ho, C_, dtype=torch.uint8, device=DEV), torch.zeros(N, Ho, Ho, C_, dtype=torch.uint8, device=DEV)
        ops.stem_pool_fwd(dtype, N, H, H, C_, yy, sc, sh, p0, a0)
        ops.stem_pool_fwd_tot(dtype, N, H, H, C_, yy, ops.bn_tot(t, R2, 4000, ga, beta=be), p1, a1)
        assert torch.equal(p1, p0) and torch.equal(a1, a0)


def test_batched_finalize_equals_the_per_layer_kernels():
    from frx import ops
    import struct
    g = torch.Generator().manual_seed(21)
    f2i = lambda v: struct.unpack("i", struct.pack("f", v))[0]
    layers, rows_f, rows_b, keep, blk = [], [], [], [], 0
    for Cc, R in ((64, 32), (256, 8), (2048, 8), (512, 8)):
        count = 1000 + Cc
        t = torch.zeros(R, 2, Cc, device=DEV)
        t[:, 0] = (torch.randn(R, Cc, generator=g) * 10).to(DEV); t[:, 1] = (torch.rand(R, Cc, generator=g) * 300 + 100).to(DEV)
        ga, be = (torch.rand(Cc, generator=g) + 0.5).to(DEV), (torch.randn(Cc, generator=g) * 0.3).to(DEV)
        rm, rv = torch.randn(Cc, generator=g).to(DEV), (torch.rand(Cc, generator=g) + 0.5).to(DEV)
        ref = [torch.zeros(Cc, device=DEV) for _ in range(4)]
        rm0, rv0 = rm.clone(), rv.clone()
        ops.bn_finalize(t, R, Cc, count, ga, be, rm0, rv0, *ref, eps=1e-5, momentum=0.1)
        out = [torch.zeros(Cc, device=DEV) for _ in range(4)]
        tf = t.clone()
        rows_f.append((tf.data_ptr(), R, Cc, count, ga.data_ptr(), be.data_ptr(), rm.data_ptr(), rv.data_ptr(), out[0].data_ptr(),
                       out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), blk, f2i(1e-5), f2i(0.1), 0))
        # backward twin on the same buffer contents
        dg0, db0, coef0 = torch.full((Cc,), 0.5, device=DEV), torch.full((Cc,), -0.25, device=DEV), torch.zeros(3 * Cc, device=DEV)
        ops.bn_bwd_finalize(t, R, Cc, count, ga, ref[0], ref[1], dg0, db0, coef0)
        dg, db, coef = torch.full((Cc,), 0.5, device=DEV), torch.full((Cc,), -0.25, device=DEV), torch.zeros(3 * Cc, device=DEV)
        tb = t.clone()
        rows_b.append((tb.data_ptr(), R, Cc, count, ga.data_ptr(), ref[0].data_ptr(), ref[1].data_ptr(), dg.data_ptr(), db.data_ptr(),
                       coef.data_ptr(), 0, 0, blk, 0, 0, 0))
        blk += (Cc + 255) // 256
        layers.append((ref, rm0, rv0, out, rm, rv, tf, dg0, db0, coef0, dg, db, coef, tb))
        keep += [t, ga, be]
    tab_f = torch.tensor(rows_f, dtype=torch.int64, device=DEV)
    tab_b = torch.tensor(rows_b, dtype=torch.int64, device=DEV)
    ops.bn_finalize_batched(tab_f, len(rows_f), blk)
    ops.bn_bwd_finalize_batched(tab_b, len(rows_b), blk)
    torch.cuda.synchronize()
    for ref, rm0, rv0, out, rm, rv, tf, dg0, db0, coef0, dg, db, coef, tb in layers:
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        assert torch.equal(rm, rm0) and torch.equal(rv, rv0)
        assert torch.equal(dg, dg0) and torch.equal(db, db0) and torch.equal(coef, coef0)
        assert float(tf.abs().max()) == 0.0 and float(tb.abs().max()) == 0.0, "the closing launch zeroes the rows"


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (128, 128, 3, 1, 14), (2048, 512, 1, 1, 4), (256, 256, 3, 2, 14)],
                         ids=lambda s: "x".join(map(str, s)))
def test_dgrad_bn_tot_equals_dgrad_bn(dtype, shape):
    """prologue coefficients from totals == from frx_bn_bwd_finalize (bit for bit); epilogue sums into totals == partial rows"""
    from frx import ops
    Ci, Co, k, stride, Hi = shape
    N, R, pad = 20, 8, k // 2
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, k, k, stride, pad)
    T = ops.TORCH_DT[dtype]
    g = torch.Generator().manual_seed(3)
    dz, y = _rand(dtype, N, d.Ho, d.Wo, Co, seed=1), _rand(dtype, N, d.Ho, d.Wo, Co, seed=2)
    wt = _rand(dtype, Ci, k, k, Co, seed=4, scale=(Ci * k * k) ** -0.5)
    count = N * d.Ho * d.Wo
    tb = torch.zeros(R, 2, Co, device=DEV)
    tb[:, 0] = (torch.randn(R, Co, generator=g) * 3).to(DEV); tb[:, 1] = (torch.randn(R, Co, generator=g) * 3).to(DEV)
    gamma = (torch.rand(Co, generator=g) + 0.5).to(DEV)
    mean, invstd = (torch.randn(Co, generator=g) * 0.2).to(DEV), (torch.rand(Co, generator=g) + 0.5).to(DEV)
    coef = torch.zeros(3 * Co, device=DEV)
    ops.bn_bwd_finalize(tb, R, Co, count, gamma, mean, invstd, None, None, coef)
    ey = _rand(dtype, N, Hi, Hi, Ci, seed=7)
    esc, esh = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    emu, eis = (torch.randn(Ci, generator=g) * 0.2).to(DEV), (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    add = _rand(dtype, N, Hi, Hi, Ci, seed=6)
    prow = ops.conv_dgrad_stat_rows(d)
    part = torch.zeros(prow, 2, Ci, device=DEV)
    dx0, dx1 = torch.empty(N, Hi, Hi, Ci, dtype=T, device=DEV), torch.empty(N, Hi, Hi, Ci, dtype=T, device=DEV)
    kw = dict(addend=add, pro_y=y, epi_y=ey, epi_scale=esc, epi_shift=esh, epi_mean=emu, epi_invstd=eis)
    side0 = torch.zeros_like(dz) if k == 1 else None
    side1 = torch.zeros_like(dz) if k == 1 else None
    ops.conv_dgrad_bn(d, dz, wt, dx0, pro_coef=coef, epi_partial=part, pro_dy_out=side0, **kw)
    tout = torch.zeros(R, 2, Ci, device=DEV)
    ops.conv_dgrad_bn(d, dz, wt, dx1, pro_tot=ops.bn_tot(tb, R, count, gamma, mean=mean, invstd=invstd), epi_totals=tout, epi_replicas=R,
                      pro_dy_out=side1, **kw)
    assert torch.equal(dx1, dx0)
    if k == 1:
        assert torch.equal(side1, side0)
    a, b = tout.sum(0), part.sum(0)
    assert (a - b).abs().max().item() <= 3e-5 * b.abs().max().item()
    # the stand-alone apply / reduce kernels
    rows = N * d.Ho * d.Wo
    dy0, dy1 = torch.empty_like(dz), torch.empty_like(dz)
    ops.bn_bwd_apply(dtype, rows, Co, dz, y, mean, invstd, coef, dy0)
    ops.bn_bwd_apply_tot(dtype, rows, Co, dz, y, ops.bn_tot(tb, R, count, gamma, mean=mean, invstd=invstd), dy1)
    assert torch.equal(dy1, dy0)
    nblk = ops.bn_bwd_partial_rows(rows, Co)
    p0, t1 = torch.zeros(nblk, 2, Co, device=DEV), torch.zeros(R, 2, Co, device=DEV)
    z0, z1 = torch.empty_like(dz), torch.empty_like(dz)
    sc2, sh2 = (torch.rand(Co, generator=g) + 0.5).to(DEV), (torch.randn(Co, generator=g) * 0.3).to(DEV)
    ops.bn_bwd_reduce(dtype, rows, Co, dz, y, mean, invstd, p0, scale=sc2, shift=sh2, relu=True, dz_out=z0)
    ops.bn_bwd_reduce_tot(dtype, rows, Co, dz, y, mean, invstd, t1, R, scale=sc2, shift=sh2, relu=True, dz_out=z1)
    assert torch.equal(z1, z0)
    assert (t1.sum(0) - p0.sum(0)).abs().max().item() <= 3e-5 * p0.sum(0).abs().max().item()


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
def test_stem_bwd_tot_equals_stem_bwd(dtype):
    from frx import ops
    N, H, Cc, R = 4, 56, 64, 32
    Ho = (H + 2 - 3) // 2 + 1
    g = torch.Generator().manual_seed(H)
    y = _rand(dtype, N, H, H, Cc, seed=1)
    scale, shift = (torch.rand(Cc, generator=g) + 0.5).to(DEV), (torch.randn(Cc, generator=g) * 0.3).to(DEV)
    mean, invstd = (torch.randn(Cc, generator=g) * 0.2).to(DEV), (torch.rand(Cc, generator=g) + 0.7).to(DEV)
    gamma = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    dout = _rand(dtype, N, Ho, Ho, Cc, seed=2)
    pooled = torch.empty(N, Ho, Ho, Cc, dtype=y.dtype, device=DEV)
    arg = torch.empty(N, Ho, Ho, Cc, dtype=torch.uint8, device=DEV)
    ops.stem_pool_fwd(dtype, N, H, H, Cc, y, scale, shift, pooled, arg)
    part = torch.zeros(ops.stem_bwd_partial_rows(), 2, Cc, device=DEV)
    ops.stem_bwd_reduce(dtype, N, H, H, Cc, dout, arg, y, scale, shift, mean, invstd, part)
    tot = torch.zeros(R, 2, Cc, device=DEV)
    ops.stem_bwd_reduce_tot(dtype, N, H, H, Cc, dout, arg, y, scale, shift, mean, invstd, tot, R)
    assert (tot.sum(0) - part.sum(0)).abs().max().item() <= 3e-5 * part.sum(0).abs().max().item()
    # apply: coefficients from the SAME R rows through both forms
    rows = N * H * H
    coef = torch.zeros(3 * Cc, device=DEV)
    ops.bn_bwd_finalize(tot, R, Cc, rows, gamma, mean, invstd, None, None, coef)
    dy0, dy1 = torch.empty_like(y), torch.empty_like(y)
    ops.stem_bwd_apply(dtype, N, H, H, Cc, dout, arg, y, scale, shift, coef, dy0)
    ops.stem_bwd_apply_tot(dtype, N, H, H, Cc, dout, arg, y, scale, shift, ops.bn_tot(tot, R, rows, gamma, mean=mean, invstd=invstd), dy1)
    assert torch.equal(dy1, dy0)
