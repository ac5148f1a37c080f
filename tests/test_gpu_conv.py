"""GPU parity of the implicit-GEMM convolution kernels (forward with fused BN+ReLU prologue and
statistics epilogue, dgrad, wgrad, stem) through the C ABI against ATen CPU fp32 -- the oracle at
the backbone boundary (torchvision itself is absent: parity unpinned upstream)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (Ci, Co, k, stride, Hi) -- the unique ResNet-50 shapes of SURVEY Appendix A, small batch
SHAPES = [
    (64, 64, 1, 1, 28), (64, 64, 3, 1, 28), (64, 256, 1, 1, 28), (256, 64, 1, 1, 28),
    (128, 128, 3, 2, 28), (256, 512, 1, 2, 28), (512, 128, 1, 1, 14), (128, 128, 3, 1, 14),
    (256, 256, 3, 2, 14), (1024, 256, 1, 1, 7), (256, 256, 3, 1, 7), (512, 512, 3, 2, 7),
    (2048, 512, 1, 1, 4), (512, 512, 3, 1, 4), (512, 2048, 1, 1, 4),
    # the remaining Appendix-A shapes (these were reached only through the whole-net test)
    (256, 128, 1, 1, 28), (128, 512, 1, 1, 14), (512, 256, 1, 1, 14), (256, 1024, 1, 1, 7), (512, 1024, 1, 2, 14),
    (1024, 512, 1, 1, 7), (1024, 2048, 1, 2, 7),
]


def _tol(dtype):
    from frx import ops
    return (2e-4, 2e-4) if dtype == ops.F32 else (2e-2, 2e-2)


def _mk(dtype, *shape, scale=1.0, seed=0):
    from frx import ops
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(*shape, generator=g) * scale).to(ops.TORCH_DT[dtype])
    return t


def _close(got, ref, dtype, what):
    rtol, atol = _tol(dtype)
    ref = ref.float()
    got = got.float().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= atol * scale + 1e-6, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _sid(s):
    return f"{s[0]}x{s[1]}k{s[2]}s{s[3]}h{s[4]}"


# The instantiated k_igemm tiles ("BMxBNxWAVESxKC", csrc/conv_launch.h).  pick_tile() chooses by the number of tiles a
# launch has, so a 3-image test only ever reaches the small ones: FRX_IGEMM_TILE (read per launch) forces each
# production tile through the same ATen comparisons, and test_conv_production_size_vs_aten runs the shapes at the
# bench's own batch, where pick_tile makes the choice itself.
TILES = ["128x128x8x64", "64x128x4x128", "128x64x4x64", "64x64x4x64"]
TILE_SHAPES = [(64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (128, 128, 3, 1, 14), (128, 128, 3, 2, 28), (256, 512, 1, 2, 28),
               (1024, 256, 1, 1, 7), (256, 256, 3, 1, 7), (512, 512, 3, 2, 7), (512, 512, 3, 1, 4), (2048, 512, 1, 1, 4)]


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=[_sid(s) for s in SHAPES])
def test_conv_fwd_dgrad_wgrad(dtype, shape):
    _conv_case(dtype, shape, 3)


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("shape", TILE_SHAPES, ids=[_sid(s) for s in TILE_SHAPES])
def test_conv_forced_tiles_vs_aten(monkeypatch, dtype, shape, tile):
    """every instantiated tile (eight-wave 128x128, 64x128 with 128-byte K-chunks, 128x64, 64x64) on forward, input
    gradient (incl. the stride-2 parity-class path) and their epilogues, batch 9: several row tiles and a ragged tail"""
    monkeypatch.setenv("FRX_IGEMM_TILE", tile)
    _conv_case(dtype, shape, 9)


def _conv_case(dtype, shape, N):
    from frx import ops
    Ci, Co, k, stride, Hi = shape
    pad = k // 2
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, k, k, stride, pad)
    x = _mk(dtype, N, Hi, Hi, Ci, seed=1)                       # NHWC
    w = _mk(dtype, Co, k, k, Ci, scale=(Ci * k * k) ** -0.5, seed=2)   # KRSC
    sc = (torch.rand(Ci, generator=torch.Generator().manual_seed(3)) + 0.5)
    sc[::3] *= -1                                               # negative gammas occur in training
    sh = torch.randn(Ci, generator=torch.Generator().manual_seed(4)) * 0.3
    xd, wd = x.to(DEV), w.to(DEV)
    # --- forward with prologue + stats
    rows = ops.conv_stat_rows(d)
    y = torch.empty(N, d.Ho, d.Wo, Co, dtype=x.dtype, device=DEV)
    part = torch.zeros(rows, 2, Co, device=DEV)
    ops.conv_fwd(d, xd, wd, y, in_scale=sc.to(DEV), in_shift=sh.to(DEV), in_relu=True, stat_partial=part)
    xin = torch.relu(x.float() * sc + sh).to(x.dtype).float()   # what the kernel stages (rounded to T)
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=stride, padding=pad)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    _close(y, ref, dtype, "conv_fwd")
    yq = y.float().cpu()
    _close(part[:, 0].sum(0), yq.sum((0, 1, 2)), 0, "stat sum")
    _close(part[:, 1].sum(0), (yq * yq).sum((0, 1, 2)), 0, "stat sumsq")
    # --- forward without prologue
    y2 = torch.empty_like(y)
    ops.conv_fwd(d, xd, wd, y2)
    ref2 = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=stride, padding=pad)
    _close(y2, ref2.permute(0, 2, 3, 1), dtype, "conv_fwd plain")
    # --- dgrad (+ addend)
    dy = _mk(dtype, N, d.Ho, d.Wo, Co, seed=5)
    wt = torch.empty(Ci, k, k, Co, dtype=x.dtype, device=DEV)
    wk = torch.empty(Co, k, k, Ci, dtype=x.dtype, device=DEV)
    ops.weight_prep(dtype, Co, k * k, Ci, w.float().to(DEV).contiguous(), krsc=wk, crsk=wt)
    assert torch.equal(wk.cpu(), w)
    assert torch.equal(wt.cpu(), w.permute(3, 1, 2, 0).contiguous())
    add = _mk(dtype, N, Hi, Hi, Ci, seed=6)
    dx = torch.empty(N, Hi, Hi, Ci, dtype=x.dtype, device=DEV)
    ops.conv_dgrad(d, dy.to(DEV), wt, dx, addend=add.to(DEV))
    refdx = torch.nn.grad.conv2d_input((N, Ci, Hi, Hi), w.float().permute(0, 3, 1, 2), dy.float().permute(0, 3, 1, 2),
                                       stride=stride, padding=pad).permute(0, 2, 3, 1) + add.float()
    _close(dx, refdx, dtype, "conv_dgrad")
    # --- wgrad with prologue (accumulates into dw)
    dw = torch.full((Co, k, k, Ci), 0.5, device=DEV)
    ops.conv_wgrad(d, xd, dy.to(DEV), dw, in_scale=sc.to(DEV), in_shift=sh.to(DEV), in_relu=True)
    refdw = torch.nn.grad.conv2d_weight(xin.permute(0, 3, 1, 2), (Co, Ci, k, k), dy.float().permute(0, 3, 1, 2),
                                        stride=stride, padding=pad).permute(0, 2, 3, 1) + 0.5
    _close(dw, refdw, dtype, "conv_wgrad")


PROD_SHAPES = [((256, 256, 3, 1, 7), (128, 128)), ((512, 512, 3, 1, 4), (64, 128)), ((1024, 256, 1, 1, 7), (64, 128)),
               ((128, 128, 3, 2, 28), (128, 128)), ((64, 256, 1, 1, 28), (128, 128)), ((256, 64, 1, 1, 28), (128, 64))]


@pytest.mark.parametrize("shape,tile", PROD_SHAPES, ids=[_sid(s) for s, _ in PROD_SHAPES])
def test_conv_production_size_vs_aten(shape, tile):
    """BASELINE configs[1]'s own launches: batch 256, bf16, the tile pick_tile() chooses at that size (asserted), forward
    with prologue and statistics, input gradient with addend, weight gradient -- all against ATen CPU fp32"""
    from frx import ops
    Ci, Co, k, stride, Hi = shape
    d = ops.conv_desc(ops.BF16, 256, Hi, Hi, Ci, Co, k, k, stride, k // 2)
    assert ops._igemm_tile(d) == tile, "pick_tile no longer chooses the tile this test was written for"
    _conv_case(ops.BF16, shape, 256)


# (N, H, W, Ci, Co): the four 3x3 / stride-1 layers at the bench's batch, then ragged cases -- a row tail, H != W, one and
# two-pixel images (every tap but the centre leaves the image), the widest image a patch buffer holds, 192 = three 64-channel
# chunks... of the OUTPUT (the forward needs an even number of 64-byte chunks of the input: 64 | Ci)
PATCH_CASES = [(256, 28, 28, 64, 64), (256, 14, 14, 128, 128), (256, 7, 7, 256, 256), (256, 4, 4, 512, 512), (3, 14, 14, 128, 128),
               (5, 9, 13, 64, 128), (2, 28, 28, 128, 256), (1, 30, 30, 64, 128), (7, 5, 3, 192, 128), (1, 1, 1, 64, 128), (1, 2, 2, 128, 64),
               (2, 11, 6, 128, 192)]


@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: "n%dh%dw%d_%dto%d" % c)
def test_patch_3x3_forward_vs_aten(case):
    """the patch-mode 3x3 kernel (k_igemm P3: BN+ReLU prologue applied once per staged patch, nine taps read at per-lane row
    offsets, weights through an LDS-DMA ring) against ATen CPU fp32 -- plain epilogue, statistics as replicated totals,
    and (where the layer's partial-rows tile is the patch tile) statistics as partial rows"""
    from frx import ops
    N, H, W, Ci, Co = case
    d = ops.conv_desc(ops.BF16, N, H, W, Ci, Co, 3, 3, 1, 1)
    assert ops.conv_patch_mode(d), "geometry no longer eligible for the kernel this test was written for"
    x = _mk(1, N, H, W, Ci, seed=1)
    w = _mk(1, Co, 3, 3, Ci, scale=(Ci * 9) ** -0.5, seed=2)
    sc = (torch.rand(Ci, generator=torch.Generator().manual_seed(3)) + 0.5)
    sc[::3] *= -1
    sh = torch.randn(Ci, generator=torch.Generator().manual_seed(4)) * 0.3
    xd, wd, scd, shd = x.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV)
    xin = torch.relu(x.float() * sc + sh).to(x.dtype).float()
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=1, padding=1).permute(0, 2, 3, 1).contiguous()
    y = torch.full((N, H, W, Co), float("nan"), dtype=x.dtype, device=DEV)
    ops.conv_fwd(d, xd, wd, y, in_scale=scd, in_shift=shd, in_relu=True)
    _close(y, ref, 1, "patch fwd, plain epilogue")
    # the same launch keeping its prologue's output (what the 3x3 weight gradient stages): every element, exactly
    xn = torch.full((N, H, W, Ci), float("nan"), dtype=x.dtype, device=DEV)
    yk = torch.full_like(y, float("nan"))
    ops.conv_fwd_keep(d, xd, wd, yk, xn, in_scale=scd, in_shift=shd, in_relu=True)
    assert torch.equal(yk, y)
    want = torch.relu(torch.addcmul(sh, x.float(), sc)).to(x.dtype)          # fma(x, scale, shift), as the kernel evaluates it
    got = xn.cpu()
    assert torch.isfinite(got.float()).all(), "kept input has unwritten elements"
    assert (got.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()
    assert (got != want).float().mean().item() < 2e-3
    # statistics into R replicated rows (the training step's form); the prologue constants given as arrays here
    R = 4
    tot = torch.zeros(R, 2, Co, device=DEV)
    y2 = torch.full_like(y, float("nan"))
    g = torch.Generator().manual_seed(5)
    gamma, beta = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    count = float(N * H * W)
    tin = torch.zeros(2, 2, Ci, device=DEV)
    xs = x.float()
    tin[0, 0] = xs.sum((0, 1, 2)).to(DEV) * 0.25; tin[1, 0] = xs.sum((0, 1, 2)).to(DEV) * 0.75
    tin[0, 1] = (xs * xs).sum((0, 1, 2)).to(DEV) * 0.5; tin[1, 1] = (xs * xs).sum((0, 1, 2)).to(DEV) * 0.5
    ops.conv_fwd_tot(d, xd, wd, y2, in_bn=ops.bn_tot(tin, 2, count, gamma, beta=beta), stat_totals=tot, stat_replicas=R)
    mean = xs.mean((0, 1, 2)); var = (xs * xs).mean((0, 1, 2)) - mean * mean
    sc2 = gamma.cpu() / torch.sqrt(var + 1e-5); sh2 = beta.cpu() - mean * sc2
    xin2 = torch.relu(xs * sc2 + sh2).to(x.dtype).float()
    ref2 = F.conv2d(xin2.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=1, padding=1).permute(0, 2, 3, 1).contiguous()
    _close(y2, ref2, 1, "patch fwd, constants from totals")
    yq = y2.float().cpu()
    _close(tot[:, 0].sum(0), yq.sum((0, 1, 2)), 0, "totals: sum")
    _close(tot[:, 1].sum(0), (yq * yq).sum((0, 1, 2)), 0, "totals: sum of squares")
    if ops._igemm_tile(d)[0] == 128:
        rows = ops.conv_stat_rows(d)
        part = torch.zeros(rows, 2, Co, device=DEV)
        y3 = torch.full_like(y, float("nan"))
        ops.conv_fwd(d, xd, wd, y3, in_scale=scd, in_shift=shd, in_relu=True, stat_partial=part)
        assert torch.equal(y3, y), "the epilogue flavour must not change the outputs"
        yq = y3.float().cpu()
        _close(part[:, 0].sum(0), yq.sum((0, 1, 2)), 0, "partial rows: sum")
        _close(part[:, 1].sum(0), (yq * yq).sum((0, 1, 2)), 0, "partial rows: sum of squares")


@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: "n%dh%dw%d_%dto%d" % c)
def test_patch_3x3_input_gradient_vs_aten(case):
    """the patch-mode input gradient: dy = alpha dz + beta y + gam (BN backward of this conv's output) staged once per patch,
    dx = conv^T(dy) masked by the ReLU of the BatchNorm below, dz and the two masked statistics -- against the same
    composition in fp32 on the CPU (ATen's conv2d_input)"""
    from frx import ops
    N, H, W, Ci, Co = case
    d = ops.conv_desc(ops.BF16, N, H, W, Ci, Co, 3, 3, 1, 1)
    if Co % 64:
        pytest.skip("the input gradient gathers Co channels: needs 64 | Co")
    assert ops.conv_patch_mode(d, True)
    T = torch.bfloat16
    dz, yraw = _mk(1, N, H, W, Co, seed=1), _mk(1, N, H, W, Co, seed=2)
    g = torch.Generator().manual_seed(3)
    coef = torch.cat([torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2, torch.randn(Co, generator=g) * 0.1])
    w = _mk(1, Co, 3, 3, Ci, scale=(Co * 9) ** -0.5, seed=4)
    ey = _mk(1, N, H, W, Ci, seed=7)
    esc = torch.rand(Ci, generator=g) + 0.5
    esc[::4] *= -1
    esh = torch.randn(Ci, generator=g) * 0.3
    emu, eis = torch.randn(Ci, generator=g) * 0.2, torch.rand(Ci, generator=g) + 0.5
    dy = (coef[:Co] * dz.float() + coef[Co:2 * Co] * yraw.float() + coef[2 * Co:]).to(T).float()
    dx = torch.nn.grad.conv2d_input((N, Ci, H, W), w.float().permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2), stride=1, padding=1).permute(0, 2, 3, 1)
    mask = (ey.float() * esc + esh) > 0
    ref = torch.where(mask, dx, torch.zeros_like(dx))
    R = 8
    tot = torch.zeros(R, 2, Ci, device=DEV)
    out = torch.full((N, H, W, Ci), float("nan"), dtype=T, device=DEV)
    wt = w.permute(3, 1, 2, 0).contiguous().to(DEV)
    dv = lambda t: t.to(DEV)
    dy_side = torch.full((N, H, W, Co), float("nan"), dtype=T, device=DEV)     # the transformed operand, kept for the weight gradient
    ops.conv_dgrad_bn(d, dv(dz), wt, out, pro_y=dv(yraw), pro_coef=dv(coef), epi_y=dv(ey), epi_scale=dv(esc), epi_shift=dv(esh),
                      epi_mean=dv(emu), epi_invstd=dv(eis), epi_totals=tot, epi_replicas=R, pro_dy_out=dy_side)
    _close(out, ref, 1, "patch dgrad: dz")
    assert torch.isfinite(dy_side.float()).all(), "side output has unwritten elements"
    # (the kernel evaluates fma(alpha, dz, fma(beta, y, gam)): one bf16 ulp where the two roundings differ)
    assert (dy_side.float().cpu() - dy).abs().max().item() <= 2.0 ** -7 * dy.abs().max().item()
    assert (dy_side.float().cpu() != dy).float().mean().item() < 2e-3
    # a masked element whose pre-activation is within rounding of 0 may flip: compare the statistics on the kernel's own dz
    oq = out.float().cpu()
    xhat = (ey.float() - emu) * eis
    _close(tot[:, 0].sum(0), oq.sum((0, 1, 2)), 0, "patch dgrad: sum dz")
    s2 = (oq * xhat).sum((0, 1, 2))
    assert (tot[:, 1].sum(0).cpu() - s2).abs().max().item() <= 2e-3 * (s2.abs().max().item() + 1e-6) + 1e-4
    if ops._igemm_tile(d, True)[0] == 128:
        rows = ops.conv_dgrad_stat_rows(d)
        part = torch.zeros(rows, 2, Ci, device=DEV)
        out2 = torch.full_like(out, float("nan"))
        ops.conv_dgrad_bn(d, dv(dz), wt, out2, pro_y=dv(yraw), pro_coef=dv(coef), epi_y=dv(ey), epi_scale=dv(esc), epi_shift=dv(esh),
                          epi_mean=dv(emu), epi_invstd=dv(eis), epi_partial=part)
        assert torch.equal(out2, out)
        _close(part[:, 0].sum(0), oq.sum((0, 1, 2)), 0, "patch dgrad: partial rows, sum dz")


@pytest.mark.parametrize("case", [(256, 14, 14, 128, 128), (256, 7, 7, 256, 256), (256, 4, 4, 512, 512), (37, 28, 28, 64, 64)],
                         ids=lambda c: "n%dh%dw%d_%dto%d" % c)
def test_patch_3x3_is_bit_reproducible_over_many_launches(case):
    """race screen for the patch-mode kernels (counted vmcnt waits, raw barriers, staging waves): 200 launches of the forward
    (outputs + kept input) and of the input gradient (outputs + dy side output) on the same operands, in a stream that also
    keeps other work in flight, must agree bit for bit with the first -- an early read of an LDS stage shows up as a rare
    differing tile, not as a failed tolerance"""
    from frx import ops
    N, H, W, Ci, Co = case
    d = ops.conv_desc(ops.BF16, N, H, W, Ci, Co, 3, 3, 1, 1)
    x, w = _mk(1, N, H, W, Ci, seed=1).to(DEV), _mk(1, Co, 3, 3, Ci, scale=(Ci * 9) ** -0.5, seed=2).to(DEV)
    wt = w.permute(3, 1, 2, 0).contiguous()
    g = torch.Generator().manual_seed(3)
    sc, sh = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    dz, yraw = _mk(1, N, H, W, Co, seed=5).to(DEV), _mk(1, N, H, W, Co, seed=6).to(DEV)
    coef = torch.randn(3, Co, generator=g).to(DEV)
    ey = _mk(1, N, H, W, Ci, seed=7).to(DEV)
    esc, esh = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    emu, eis = (torch.randn(Ci, generator=g) * 0.2).to(DEV), (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)

    def fwd():
        y = torch.empty(N, H, W, Co, dtype=torch.bfloat16, device=DEV)
        xn = torch.empty_like(x)
        ops.conv_fwd_keep(d, x, w, y, xn, in_scale=sc, in_shift=sh, in_relu=True)
        return y, xn

    def bwd():
        dx = torch.empty_like(x)
        dy = torch.empty_like(dz)
        tot = torch.zeros(8, 2, Ci, device=DEV)
        ops.conv_dgrad_bn(d, dz, wt, dx, pro_y=yraw, pro_coef=coef, pro_dy_out=dy, epi_y=ey, epi_scale=esc, epi_shift=esh, epi_mean=emu,
                          epi_invstd=eis, epi_totals=tot, epi_replicas=8)
        return dx, dy

    for fn in (fwd, bwd):
        first = [t.clone() for t in fn()]
        bad = 0
        for i in range(200):
            if i % 4 == 0:
                noise.fill_(i & 255)                   # (memory traffic of another kernel in the same stream's wake)
            out = fn()
            bad += sum(int(not torch.equal(a, b)) for a, b in zip(out, first))
        assert bad == 0, f"{fn.__name__}: {bad} of 400 result tensors differ from the first launch"


def test_patch_3x3_matches_chunk_per_tap(monkeypatch):
    """the two main loops of the same launch (FRX_CONV3X3=0: one gathered K-chunk per tap) differ only in the order of the
    fp32 accumulation: outputs agree to bf16 rounding of equal sums, statistics to 1e-5"""
    from frx import ops
    N, H, W, Ci, Co = 64, 14, 14, 128, 128
    d = ops.conv_desc(ops.BF16, N, H, W, Ci, Co, 3, 3, 1, 1)
    x, w = _mk(1, N, H, W, Ci, seed=1).to(DEV), _mk(1, Co, 3, 3, Ci, scale=(Ci * 9) ** -0.5, seed=2).to(DEV)
    g = torch.Generator().manual_seed(3)
    sc, sh = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    res = {}
    for v in ("0", "1"):
        monkeypatch.setenv("FRX_CONV3X3", v)
        y = torch.empty(N, H, W, Co, dtype=torch.bfloat16, device=DEV)
        rows = ops.conv_stat_rows(d)
        part = torch.zeros(rows, 2, Co, device=DEV)
        ops.conv_fwd(d, x, w, y, in_scale=sc, in_shift=sh, in_relu=True, stat_partial=part)
        res[v] = (y.float(), part.sum(0))
    rel = ((res["1"][0] - res["0"][0]).norm() / res["0"][0].norm()).item()
    assert rel < 2e-4, rel
    assert ((res["1"][1] - res["0"][1]).abs().max() / res["0"][1].abs().max()).item() < 1e-5


@pytest.mark.parametrize("shape", [(1024, 256, 1, 1, 7), (256, 256, 3, 1, 7), (256, 64, 1, 1, 28), (256, 256, 3, 2, 14)], ids=_sid)
@pytest.mark.parametrize("merge_mask", [False, True], ids=["relu_bn_mask", "merge_mask"])
def test_fused_bn_backward_production_size(shape, merge_mask):
    """the fused BN-backward input / weight gradients at batch 256, bf16, on the tiles the bench runs them on"""
    from frx import ops
    _fused_bn_case(ops.BF16, shape, merge_mask, 256)


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
def test_stem(dtype):
    from frx import ops
    N, H = 3, 112
    g = torch.Generator().manual_seed(0)
    img = torch.rand(N, 3, H, H, generator=g) * 2 - 1
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05             # torchvision layout
    hp, wp = ops.stem_padded_dims(H, H)
    xin = torch.empty(N, hp, wp, 4, dtype=ops.TORCH_DT[dtype], device=DEV)
    ops.input_prep(dtype, img.to(DEV), xin)
    # uint8 path must give the same tensor as ToTensor+Normalize on the CPU
    u8 = (torch.rand(N, H, H, 3, generator=g) * 255).to(torch.uint8)
    xin8 = torch.empty_like(xin)
    ops.input_prep(dtype, u8.to(DEV), xin8)
    ref8 = ((u8.float() / 255 - 0.5) / 0.5).to(ops.TORCH_DT[dtype])
    assert torch.equal(xin8[:, 3:3 + H, 3:3 + H, :3].cpu(), ref8)
    assert xin8[:, :3].abs().sum() == 0 and xin8[..., 3].abs().sum() == 0
    wk = torch.zeros(64, 7, 8, 4)
    wk[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    wk = wk.to(ops.TORCH_DT[dtype])
    d = ops.conv_desc(dtype, N, H, H, 3, 64, 7, 7, 2, 3, stem=True)
    rows = ops.conv_stat_rows(d)
    y = torch.empty(N, 56, 56, 64, dtype=xin.dtype, device=DEV)
    part = torch.zeros(rows, 2, 64, device=DEV)
    ops.conv_fwd(d, xin, wk.to(DEV), y, stat_partial=part)
    imgq = img.to(ops.TORCH_DT[dtype]).float()
    ref = F.conv2d(imgq, wk[:, :, :7, :3].float().permute(0, 3, 1, 2), stride=2, padding=3).permute(0, 2, 3, 1)
    _close(y, ref, dtype, "stem fwd")
    _close(part[:, 0].sum(0), y.float().cpu().sum((0, 1, 2)), 0, "stem stat")
    dy = _mk(dtype, N, 56, 56, 64, seed=9)
    dw = torch.zeros(64, 7, 8, 4, device=DEV)
    ops.conv_wgrad(d, xin, dy.to(DEV), dw)
    refdw = torch.nn.grad.conv2d_weight(imgq, (64, 3, 7, 7), dy.float().permute(0, 3, 1, 2), stride=2, padding=3)
    _close(dw[:, :, :7, :3], refdw.permute(0, 2, 3, 1), dtype, "stem wgrad")
    assert dw[:, :, 7, :].abs().max().item() == 0 and dw[..., 3].abs().max().item() == 0, "padding slots must stay untouched"


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
def test_fc_as_conv(dtype):
    """avgpool + Linear(2048, 512) with fp32 output and bias; ragged batch (N not a tile multiple)."""
    from frx import ops
    N = 37
    x = _mk(dtype, N, 16, 2048, seed=1)
    w = _mk(dtype, 512, 2048, scale=2048 ** -0.5, seed=2)
    b = torch.randn(512, generator=torch.Generator().manual_seed(3))
    pooled = torch.empty(N, 2048, dtype=x.dtype, device=DEV)
    ops.avgpool_fwd(dtype, N, 16, 2048, x.to(DEV), pooled)
    _close(pooled, x.float().mean(1), dtype, "avgpool")
    d = ops.conv_desc(dtype, N, 1, 1, 2048, 512, 1, 1, 1, 0)
    y = torch.empty(N, 512, device=DEV)
    ops.conv_fwd(d, pooled, w.to(DEV), y, bias=b.to(DEV), out_f32=True)
    ref = pooled.float().cpu() @ w.float().t() + b
    _close(y, ref, dtype, "fc")
    dpool = _mk(dtype, N, 2048, seed=4)
    dx = torch.empty(N, 16, 2048, dtype=x.dtype, device=DEV)
    ops.avgpool_bwd(dtype, N, 16, 2048, dpool.to(DEV), dx)
    _close(dx, (dpool.float() / 16)[:, None, :].expand(N, 16, 2048), dtype, "avgpool bwd")


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (512, 128, 1, 1, 14), (256, 512, 1, 2, 28),
                                   (128, 128, 3, 1, 14), (2048, 512, 1, 1, 4), (256, 256, 3, 2, 14), (512, 512, 3, 2, 7)],
                         ids=lambda s: f"{s[0]}x{s[1]}k{s[2]}s{s[3]}h{s[4]}")
@pytest.mark.parametrize("merge_mask", [False, True], ids=["relu_bn_mask", "merge_mask"])
def test_fused_bn_backward_in_dgrad_wgrad(dtype, shape, merge_mask):
    """frx_conv_dgrad_bn / frx_conv_wgrad_bn == (bn_bwd_apply -> dgrad -> bn_bwd_reduce) and (apply -> wgrad)."""
    _fused_bn_case(dtype, shape, merge_mask, 3)


FUSED_TILE_SHAPES = [(64, 256, 1, 1, 28), (256, 64, 1, 1, 28), (512, 128, 1, 1, 14), (128, 128, 3, 1, 14), (2048, 512, 1, 1, 4),
                     (256, 256, 3, 2, 14), (1024, 256, 1, 1, 7)]


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("shape", FUSED_TILE_SHAPES, ids=[_sid(s) for s in FUSED_TILE_SHAPES])
@pytest.mark.parametrize("merge_mask", [False, True], ids=["relu_bn_mask", "merge_mask"])
def test_fused_bn_backward_forced_tiles(monkeypatch, dtype, shape, merge_mask, tile):
    """the BN-backward prologue (PRO = 2, with the dy side store), the masking / reducing epilogues (EPI_BNBWD,
    EPI_BNBWD_OUT with byte masks) and the addend on every instantiated tile.  The reference composition below runs
    under the same forced tile; its plain dgrad is itself held to ATen by test_conv_forced_tiles_vs_aten."""
    monkeypatch.setenv("FRX_IGEMM_TILE", tile)
    _fused_bn_case(dtype, shape, merge_mask, 9)


def _fused_bn_case(dtype, shape, merge_mask, N):
    from frx import ops
    Ci, Co, k, stride, Hi = shape
    pad = k // 2
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, k, k, stride, pad)
    T = ops.TORCH_DT[dtype]
    dz = _mk(dtype, N, d.Ho, d.Wo, Co, seed=1).to(DEV)           # masked upstream gradient of THIS conv's BN
    y = _mk(dtype, N, d.Ho, d.Wo, Co, seed=2).to(DEV)            # this conv's raw output
    g = torch.Generator().manual_seed(3)
    coef = torch.cat([torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2, torch.randn(Co, generator=g) * 0.1]).to(DEV)
    w = _mk(dtype, Co, k, k, Ci, scale=(Ci * k * k) ** -0.5, seed=4).to(DEV)
    wt = w.permute(3, 1, 2, 0).contiguous()
    x = _mk(dtype, N, Hi, Hi, Ci, seed=5).to(DEV)                # conv input activation (post-ReLU) / block input
    add = _mk(dtype, N, Hi, Hi, Ci, seed=6).to(DEV)
    # reference composition with the stand-alone kernels
    dy = (coef[:Co] * dz.float() + coef[Co:2 * Co] * y.float() + coef[2 * Co:]).to(T)
    dx_ref = torch.empty(N, Hi, Hi, Ci, dtype=T, device=DEV)
    ops.conv_dgrad(d, dy, wt, dx_ref, addend=add)
    # the layer that produced this conv's input: raw output ey, BN stats, (optionally) a merge output
    ey = _mk(dtype, N, Hi, Hi, Ci, seed=7).to(DEV)
    esc = (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    esc[::4] *= -1
    esh = (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    emu, eis = (torch.randn(Ci, generator=g) * 0.2).to(DEV), (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    eout = torch.relu(_mk(dtype, N, Hi, Hi, Ci, seed=8)).to(DEV) if merge_mask else None
    rows = N * Hi * Hi
    nblk = ops.bn_bwd_partial_rows(rows, Ci)
    part_ref = torch.zeros(nblk, 2, Ci, device=DEV)
    dz_ref = torch.empty_like(dx_ref)
    ops.bn_bwd_reduce(dtype, rows, Ci, dx_ref, ey, emu, eis, part_ref, out=eout, scale=esc, shift=esh, relu=True, dz_out=dz_ref)
    # fused
    prow = ops.conv_dgrad_stat_rows(d)
    part = torch.zeros(prow, 2, Ci, device=DEV)
    dz_out = torch.empty_like(dx_ref)
    dy_side = torch.full_like(dy, float("nan")) if k == 1 else None      # 1x1: the prologue's dy is kept for the wgrad
    ops.conv_dgrad_bn(d, dz, wt, dz_out, addend=add, pro_y=y, pro_coef=coef, epi_y=ey, epi_out=eout, epi_scale=esc,
                      epi_shift=esh, epi_mean=emu, epi_invstd=eis, epi_partial=part, pro_dy_out=dy_side)
    _close(dz_out, dz_ref.float().cpu(), dtype, "fused dz")
    if merge_mask:      # the same mask as one byte per 16-byte channel group (what frx_block_merge_fwd_mask writes)
        V = 8 if dtype == 1 else 4
        bits = ((eout.float() > 0).view(-1, V).to(torch.int32) << torch.arange(V, device=DEV, dtype=torch.int32)).sum(1).to(torch.uint8)
        part_b = torch.zeros(prow, 2, Ci, device=DEV)
        dz_b = torch.empty_like(dx_ref)
        ops.conv_dgrad_bn(d, dz, wt, dz_b, addend=add, pro_y=y, pro_coef=coef, epi_y=ey, epi_out_bits=bits, epi_scale=esc,
                          epi_shift=esh, epi_mean=emu, epi_invstd=eis, epi_partial=part_b)
        assert torch.equal(dz_b, dz_out) and torch.equal(part_b, part), "bit mask and bf16 mask must give identical results"
    if dy_side is not None:
        assert torch.isfinite(dy_side.float()).all(), "side output has unwritten elements"
        _close(dy_side, dy.float().cpu(), dtype, "dy side output")
    s_ref, s = part_ref.sum(0).cpu(), part.sum(0).cpu()
    scale = s_ref.abs().max().item() + 1e-6
    assert (s - s_ref).abs().max().item() < (2e-3 if dtype == 0 else 3e-2) * scale
    # fused wgrad
    sc = (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    dw_ref = torch.zeros(Co, k, k, Ci, device=DEV)
    ops.conv_wgrad(d, x, dy, dw_ref, in_scale=sc, in_shift=sh, in_relu=True)
    dw = torch.zeros_like(dw_ref)
    ops.conv_wgrad_bn(d, x, dz, y, coef, dw, in_scale=sc, in_shift=sh, in_relu=True)
    _close(dw, dw_ref.cpu(), dtype, "fused wgrad")


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(64, 64, 1, 28), (64, 256, 1, 28), (64, 64, 3, 28)], ids=lambda s: f"{s[0]}x{s[1]}k{s[2]}")
def test_conv_many_coresident_blocks(dtype, shape):
    """Batch 64: thousands of short blocks, several per CU.  Guards a hazard met on gfx950 / ROCm 7.2: a
    buffer_store_dwordx4 with an SGPR soffset whose data registers hipcc re-used in the next instruction
    corrupted sporadic dwords only when blocks were co-resident (small-batch tests never saw it)."""
    from frx import ops
    Ci, Co, k, Hi = shape
    N = 64
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, k, k, 1, k // 2)
    T = ops.TORCH_DT[dtype]
    x, w = _mk(dtype, N, Hi, Hi, Ci, seed=1), _mk(dtype, Co, k, k, Ci, scale=0.1, seed=2)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=k // 2).permute(0, 2, 3, 1)
    xd, wd = x.to(DEV), w.to(DEV)
    one, zero = torch.ones(Ci, device=DEV), torch.zeros(Ci, device=DEV)
    for stats in (False, True):
        for pro in (False, True):
            y = torch.zeros(N, Hi, Hi, Co, dtype=T, device=DEV)
            part = torch.zeros(ops.conv_stat_rows(d), 2, Co, device=DEV) if stats else None
            kw = dict(in_scale=one, in_shift=zero, in_relu=False) if pro else {}
            ops.conv_fwd(d, xd, wd, y, stat_partial=part, **kw)
            err = (y.float().cpu() - ref).abs()
            bad = int((~(err < 0.1 + 0.02 * ref.abs())).sum())
            assert bad == 0, f"stats={stats} pro={pro}: {bad} corrupted outputs"
            if stats:
                _close(part[:, 0].sum(0), y.float().cpu().sum((0, 1, 2)), 0, "stat sum")


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(256, 128, 28), (512, 256, 14), (64, 64, 9)], ids=lambda s: f"{s[0]}x{s[1]}h{s[2]}")
@pytest.mark.parametrize("tile", [None] + TILES)
def test_dgrad_compact_stride2_addend(monkeypatch, dtype, shape, tile):
    """addend_stride = 2: the compact [N,ceil(H/2),ceil(W/2),Ci] gradient of a stride-2 1x1 branch is added at the even
    pixels -- identical to adding its zero-filled full-size form (odd H covers the ceil)."""
    from frx import ops
    Ci, Co, Hi = shape
    N = 3 if tile is None else 9
    if tile is not None:
        monkeypatch.setenv("FRX_IGEMM_TILE", tile)
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    T = ops.TORCH_DT[dtype]
    dz = _mk(dtype, N, Hi, Hi, Co, seed=1).to(DEV)
    y = _mk(dtype, N, Hi, Hi, Co, seed=2).to(DEV)
    g = torch.Generator().manual_seed(3)
    coef = torch.cat([torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2, torch.randn(Co, generator=g) * 0.1]).to(DEV)
    wt = _mk(dtype, Ci, 1, 1, Co, scale=Co ** -0.5, seed=4).to(DEV)
    Hc = (Hi + 1) // 2
    compact = _mk(dtype, N, Hc, Hc, Ci, seed=5).to(DEV)
    full = torch.zeros(N, Hi, Hi, Ci, dtype=T, device=DEV)
    full[:, ::2, ::2, :] = compact
    ref = torch.empty(N, Hi, Hi, Ci, dtype=T, device=DEV)
    ops.conv_dgrad_bn(d, dz, wt, ref, addend=full, pro_y=y, pro_coef=coef)
    out = torch.empty_like(ref)
    ops.conv_dgrad_bn(d, dz, wt, out, addend=compact, pro_y=y, pro_coef=coef, addend_stride=2)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("case", [(256, 64, 28, 6, 1), (512, 128, 14, 9, 1), (1024, 256, 7, 21, 1), (2048, 512, 4, 37, 1),
                                  (256, 128, 28, 5, 2), (512, 256, 14, 7, 2), (1024, 512, 7, 13, 2), (1024, 256, 7, 5, 0),
                                  # the bench's own batch: a dozen items per block -- the column-tile-major walk over four resident row
                                  # blocks (512 <- 128), register-held column tiles over two dozen items (256 <- 64)
                                  (512, 128, 14, 256, 1), (256, 64, 28, 250, 1)],
                         ids=lambda c: f"{c[0]}x{c[1]}h{c[2]}n{c[3]}add{c[4]}")
def test_row_resident_conv1_input_gradient_equals_the_tiled_kernel(monkeypatch, case):
    """csrc/pw_rows.hip (the conv1-type input gradient with a row block's BN-backward operand resident in LDS, LDS-DMA
    prefetched epilogue operands, statistics closed as invstd * (sum dz*y - mean * sum dz)) against k_igemm on the same
    call (FRX_PW_ROWS=0): the gradient and the kept dy bit for bit, the totals to fp32 rounding.  Ragged pixel counts
    (M % 64 != 0), the compact stride-2 addend of a block's first conv1, no addend, coefficients from totals or given."""
    from frx import ops
    Ci, Co, Hi, N, addk = case
    dtype, R = 1, 8
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    T = ops.TORCH_DT[dtype]
    g = torch.Generator().manual_seed(Ci + Hi)
    dz, y = _mk(dtype, N, Hi, Hi, Co, seed=1).to(DEV), _mk(dtype, N, Hi, Hi, Co, seed=2).to(DEV) + 0.3
    wt = _mk(dtype, Ci, 1, 1, Co, scale=Co ** -0.5, seed=4).to(DEV)
    count = N * Hi * Hi
    tb = torch.zeros(R, 2, Co, device=DEV)
    tb[:, 0] = (torch.randn(R, Co, generator=g) * 3).to(DEV); tb[:, 1] = (torch.randn(R, Co, generator=g) * 3).to(DEV)
    gamma = (torch.rand(Co, generator=g) + 0.5).to(DEV)
    mean, invstd = (torch.randn(Co, generator=g) * 0.2).to(DEV), (torch.rand(Co, generator=g) + 0.5).to(DEV)
    coef = torch.zeros(3 * Co, device=DEV)
    ops.bn_bwd_finalize(tb, R, Co, count, gamma, mean, invstd, None, None, coef)
    ey = _mk(dtype, N, Hi, Hi, Ci, seed=7).to(DEV) + 0.5
    emu, eis = (torch.randn(Ci, generator=g) * 0.2 + 0.5).to(DEV), (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    eout = torch.relu(_mk(dtype, N, Hi, Hi, Ci, seed=8)).to(DEV)
    bits = ((eout.float() > 0).view(-1, 8).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(1).to(torch.uint8)
    Hc = (Hi + 1) // 2
    add = None if addk == 0 else (_mk(dtype, N, Hi, Hi, Ci, seed=6) if addk == 1 else _mk(dtype, N, Hc, Hc, Ci, seed=6)).to(DEV)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FRX_PW_ROWS", mode)
        for given in (False, True):
            dx = torch.full((N, Hi, Hi, Ci), float("nan"), dtype=T, device=DEV)
            side = torch.full_like(dz, float("nan"))
            tout = torch.zeros(R, 2, Ci, device=DEV)
            kw = dict(pro_coef=coef) if given else dict(pro_tot=ops.bn_tot(tb, R, count, gamma, mean=mean, invstd=invstd))
            ops.conv_dgrad_bn(d, dz, wt, dx, addend=add, pro_y=y, epi_y=ey, epi_out_bits=bits, epi_mean=emu, epi_invstd=eis,
                              epi_totals=tout, epi_replicas=R, pro_dy_out=side, addend_stride=2 if addk == 2 else 0, **kw)
            f = (ctypes.c_int * 12)()
            ops._lib.lib().frx_last_conv_launch(f)
            assert f[10] == (2 if mode == "1" and Co <= 256 else 0), "which kernel ran (middle widths up to 256 take the row-resident one)"
            res[mode, given] = (dx, side, tout.sum(0))
    for given in (False, True):
        (dx0, s0, t0), (dx1, s1, t1) = res["0", given], res["1", given]
        assert torch.isfinite(dx1.float()).all() and torch.isfinite(s1.float()).all()
        assert torch.equal(dx1, dx0) and torch.equal(s1, s0)
        assert (t1 - t0).abs().max().item() <= 1e-4 * t0.abs().max().item(), (t1 - t0).abs().max().item() / t0.abs().max().item()
    assert torch.equal(res["1", False][0], res["1", True][0])


@pytest.mark.parametrize("case", [(64, 256, 28, 6), (128, 512, 14, 9), (256, 1024, 7, 21), (64, 512, 9, 5), (128, 256, 14, 3)],
                         ids=lambda c: f"{c[0]}x{c[1]}h{c[2]}n{c[3]}")
def test_row_resident_conv3_forward_equals_the_tiled_kernel(monkeypatch, case):
    """csrc/pw_rows.hip, forward flavour (the conv3-type 1x1 behind bn2 + ReLU, statistics into replicated totals) against
    k_igemm on the same call (FRX_PW_ROWS=0): the output bit for bit, the totals to fp32 summation order.  Ragged pixel
    counts; two, four and eight column tiles (two: weights and statistics held in registers)."""
    from frx import ops
    Ci, Co, Hi, N = case
    dtype, R = 1, 8
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    T = ops.TORCH_DT[dtype]
    g = torch.Generator().manual_seed(Ci + Hi)
    x = _mk(dtype, N, Hi, Hi, Ci, seed=1).to(DEV) + 0.2
    w = _mk(dtype, Co, 1, 1, Ci, scale=Ci ** -0.5, seed=4).to(DEV)
    count = N * Hi * Hi
    xf = x.float().view(-1, Ci)
    tin = torch.zeros(R, 2, Ci, device=DEV)
    tin[0, 0] = xf.sum(0); tin[0, 1] = (xf * xf).sum(0)
    tin[1:, 0] = (torch.randn(R - 1, Ci, generator=g) * 0.01).to(DEV); tin[0, 0] -= tin[1:, 0].sum(0)
    gamma, beta = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.2).to(DEV)
    res = {}
    monkeypatch.setenv("FRX_PWR_FWD256", "1")       # (the 256-channel instantiation is built but not the default: slower than k_igemm)
    for mode in ("0", "3"):
        monkeypatch.setenv("FRX_PW_ROWS", mode)
        y = torch.full((N, Hi, Hi, Co), float("nan"), dtype=T, device=DEV)
        tout = torch.zeros(R, 2, Co, device=DEV)
        ops.conv_fwd_tot(d, x, w, y, in_bn=ops.bn_tot(tin, R, count, gamma, beta=beta), in_relu=True, stat_totals=tout, stat_replicas=R)
        f = (ctypes.c_int * 12)()
        ops._lib.lib().frx_last_conv_launch(f)
        assert f[10] == (2 if mode == "3" else 0), "which kernel ran"
        res[mode] = (y, tout.sum(0))
    (y0, t0), (y1, t1) = res["0"], res["3"]
    assert torch.isfinite(y1.float()).all()
    assert torch.equal(y1, y0)
    assert (t1 - t0).abs().max().item() <= 2e-5 * t0.abs().max().item(), (t1 - t0).abs().max().item() / t0.abs().max().item()
    yf = y1.float().view(-1, Co)
    ref = torch.stack([yf.sum(0), (yf * yf).sum(0)])
    assert (t1 - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("case", [(256, 64, 28, 6), (512, 128, 14, 9), (256, 128, 28, 3), (256, 64, 9, 300)], ids=lambda c: f"{c[0]}x{c[1]}h{c[2]}n{c[3]}")
def test_streamed_conv1_forward_equals_the_tiled_kernel(monkeypatch, case):
    """csrc/pw_stream.hip, forward flavour (a block's conv1 on its stored input: operand fragments straight from global memory,
    weights resident in LDS, statistics into replicated totals) against k_igemm on the same call (FRX_PW_STREAM=0)."""
    from frx import ops
    Ci, Co, Hi, N = case
    dtype, R = 1, 8
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    T = ops.TORCH_DT[dtype]
    x = _mk(dtype, N, Hi, Hi, Ci, seed=1).to(DEV)
    w = _mk(dtype, Co, 1, 1, Ci, scale=Ci ** -0.5, seed=4).to(DEV)
    res = {}
    for mode in ("0", "2"):         # (2: every instantiated shape, also the layer2 ones the default leaves to k_igemm)
        monkeypatch.setenv("FRX_PW_STREAM", mode)
        y = torch.full((N, Hi, Hi, Co), float("nan"), dtype=T, device=DEV)
        tout = torch.zeros(R, 2, Co, device=DEV)
        ops.conv_fwd_tot(d, x, w, y, in_bn=None, stat_totals=tout, stat_replicas=R)
        f = (ctypes.c_int * 12)()
        ops._lib.lib().frx_last_conv_launch(f)
        assert f[10] == (3 if mode == "2" else 0), "which kernel ran"
        res[mode] = (y, tout.sum(0))
    (y0, t0), (y1, t1) = res["0"], res["2"]
    assert torch.isfinite(y1.float()).all() and torch.equal(y1, y0)
    assert (t1 - t0).abs().max().item() <= 2e-5 * t0.abs().max().item(), (t1 - t0).abs().max().item() / t0.abs().max().item()


@pytest.mark.parametrize("case", [(64, 256, 28, 6), (128, 512, 14, 9), (64, 256, 9, 300)], ids=lambda c: f"{c[0]}x{c[1]}h{c[2]}n{c[3]}")
def test_streamed_conv3_input_gradient_equals_the_tiled_kernel(monkeypatch, case):
    """csrc/pw_stream.hip, input-gradient flavour (conv3 of layer1 / layer2: the BN-backward operand of two full-width tensors
    built in registers, mask and statistics of bn2 behind it) against k_igemm on the same call: gradient bit for bit, totals to
    fp32 rounding (sum dz * xhat is closed as invstd * (sum dz * y - mean * sum dz)).  Coefficients from totals or given."""
    from frx import ops
    Ci, Co, Hi, N = case
    dtype, R = 1, 8
    d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, 1, 1, 1, 0)
    T = ops.TORCH_DT[dtype]
    g = torch.Generator().manual_seed(Ci + Hi)
    dz, y = _mk(dtype, N, Hi, Hi, Co, seed=1).to(DEV), _mk(dtype, N, Hi, Hi, Co, seed=2).to(DEV) + 0.3
    wt = _mk(dtype, Ci, 1, 1, Co, scale=Co ** -0.5, seed=4).to(DEV)
    count = N * Hi * Hi
    tb = torch.zeros(R, 2, Co, device=DEV)
    tb[:, 0] = (torch.randn(R, Co, generator=g) * 3).to(DEV); tb[:, 1] = (torch.randn(R, Co, generator=g) * 3).to(DEV)
    gamma = (torch.rand(Co, generator=g) + 0.5).to(DEV)
    mean, invstd = (torch.randn(Co, generator=g) * 0.2).to(DEV), (torch.rand(Co, generator=g) + 0.5).to(DEV)
    coef = torch.zeros(3 * Co, device=DEV)
    ops.bn_bwd_finalize(tb, R, Co, count, gamma, mean, invstd, None, None, coef)
    ey = _mk(dtype, N, Hi, Hi, Ci, seed=7).to(DEV) + 0.5
    esc = (torch.rand(Ci, generator=g) + 0.5).to(DEV); esc[::4] *= -1
    esh = (torch.randn(Ci, generator=g) * 0.3).to(DEV)
    emu, eis = (torch.randn(Ci, generator=g) * 0.2 + 0.5).to(DEV), (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("FRX_PW_STREAM", mode)
        for given in (False, True):
            dx = torch.full((N, Hi, Hi, Ci), float("nan"), dtype=T, device=DEV)
            tout = torch.zeros(R, 2, Ci, device=DEV)
            kw = dict(pro_coef=coef) if given else dict(pro_tot=ops.bn_tot(tb, R, count, gamma, mean=mean, invstd=invstd))
            ops.conv_dgrad_bn(d, dz, wt, dx, pro_y=y, epi_y=ey, epi_scale=esc, epi_shift=esh, epi_mean=emu, epi_invstd=eis,
                              epi_totals=tout, epi_replicas=R, **kw)
            f = (ctypes.c_int * 12)()
            ops._lib.lib().frx_last_conv_launch(f)
            assert f[10] == (3 if mode == "2" else 0), "which kernel ran"
            res[mode, given] = (dx, tout.sum(0))
    for given in (False, True):
        (dx0, t0), (dx1, t1) = res["0", given], res["2", given]
        assert torch.isfinite(dx1.float()).all() and torch.equal(dx1, dx0)
        assert (t1 - t0).abs().max().item() <= 1e-4 * t0.abs().max().item(), (t1 - t0).abs().max().item() / t0.abs().max().item()


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
def test_grouped_wgrad_matches_per_layer(dtype):
    """frx_wgrad_group_*: one persistent launch over a work list == the per-layer frx_conv_wgrad / frx_conv_wgrad_bn
    calls (every kernel variant: 1x1 / 3x3 / stride 2 / stem, with and without the two prologues, both tile sizes)."""
    from frx import ops
    N = 5
    T = ops.TORCH_DT[dtype]
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 256, 1, 1, 14, True, True), (256, 64, 1, 1, 14, False, False), (128, 128, 3, 1, 14, True, False),
              (128, 128, 3, 2, 14, True, False), (256, 512, 1, 2, 14, False, True), (64, 64, 3, 1, 28, False, False),
              (512, 128, 1, 1, 7, True, False), (64, 64, 1, 1, 28, True, True)]
    jobs, refs = [], []
    for (Ci, Co, k, st, Hi, pro, ypro) in shapes:
        d = ops.conv_desc(dtype, N, Hi, Hi, Ci, Co, k, k, st, k // 2)
        x = _mk(dtype, N, Hi, Hi, Ci, seed=Ci + Co).to(DEV)
        dy = _mk(dtype, N, d.Ho, d.Wo, Co, seed=Ci + 1).to(DEV)
        y = _mk(dtype, N, d.Ho, d.Wo, Co, seed=Ci + 2).to(DEV)
        sc, sh = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
        coef = torch.cat([torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2, torch.randn(Co, generator=g) * 0.1]).to(DEV)
        kw = dict(in_scale=sc, in_shift=sh, in_relu=True) if pro else {}
        ref = torch.zeros(Co, k, k, Ci, device=DEV)
        if ypro and k == 1:
            ops.conv_wgrad_bn(d, x, dy, y, coef, ref, **kw)
            jobs.append(dict(d=d, x=x, dy=dy, pro_y=y, pro_coef=coef, dw=torch.zeros_like(ref), **kw))
        else:
            ops.conv_wgrad(d, x, dy, ref, **kw)
            jobs.append(dict(d=d, x=x, dy=dy, dw=torch.zeros_like(ref), **kw))
        refs.append(ref)
    H = 16
    hp, wp = ops.stem_padded_dims(H, H)
    ds = ops.conv_desc(dtype, N, H, H, 3, 64, 7, 7, 2, 3, stem=True)
    xin = torch.zeros(N, hp, wp, 4, dtype=T, device=DEV)
    xin[:, 3:3 + H, 3:3 + H, :3] = _mk(dtype, N, H, H, 3, seed=77).to(DEV)
    dys = _mk(dtype, N, ds.Ho, ds.Wo, 64, seed=78).to(DEV)
    ref = torch.zeros(64, 7, 8, 4, device=DEV)
    ops.conv_wgrad(ds, xin, dys, ref)
    jobs.append(dict(d=ds, x=xin, dy=dys, dw=torch.zeros_like(ref)))
    refs.append(ref)
    grp = ops.wgrad_group_plan(dtype, jobs)
    ops.wgrad_group_run(grp)
    ops.wgrad_group_run(grp)            # accumulates like the per-layer calls
    for j, r in zip(jobs, refs):
        _close(j["dw"] * 0.5, r.cpu(), dtype, "grouped wgrad %s" % (tuple(r.shape),))


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
def test_decomposed_wgrad_equals_the_two_tensor_form(dtype):
    """Round 4: the weight gradient of a 1x1 conv whose dy is the BatchNorm backward alpha*dz + beta*y + gam, WITHOUT reading y
    (frx_wgrad_job.gram / xsum + frx_wgrad_gram_finish): dW = alpha (.) dz^T x + beta (.) W (x^T x) + gam (x) sum(x), because
    y = x W^T.  Against the two-tensor form (pro_y / pro_coef) of the same list, and both against a float64 evaluation of
    sum_m dy[m, co] x[m, ci] on the same tensors: the decomposed form must be at least as close to float64 as the form it
    replaces.  y is the library's own forward of x (bf16: rounded as the training step stores it).  Shapes: layer1 / layer2
    conv3 and layer1's projection at reduced batch, a ragged pixel count, both tile sizes, with and without the prologue."""
    from frx import ops
    T = ops.TORCH_DT[dtype]
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 256, 28, 6, True), (128, 512, 14, 9, True), (64, 256, 28, 4, False), (64, 128, 9, 3, True), (128, 128, 5, 7, False)]
    two, dec, refs = [], [], []
    for (Ci, Co, H, N, pro) in shapes:
        d = ops.conv_desc(dtype, N, H, H, Ci, Co, 1, 1, 1, 0)
        x = _mk(dtype, N, H, H, Ci, seed=Ci + Co + H).to(DEV)
        w = (_mk(dtype, Co, 1, 1, Ci, seed=Co) * Ci ** -0.5).to(T).to(DEV)
        sc, sh = (torch.rand(Ci, generator=g) + 0.5).to(DEV), (torch.randn(Ci, generator=g) * 0.3).to(DEV)
        kw = dict(in_scale=sc, in_shift=sh, in_relu=True) if pro else {}
        y = torch.empty(N, H, H, Co, dtype=T, device=DEV)
        ops.conv_fwd(d, x, w, y, **kw)
        dz = _mk(dtype, N, H, H, Co, seed=Ci + 3).to(DEV)
        coef = torch.cat([torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2, torch.randn(Co, generator=g) * 0.1]).to(DEV)
        two.append(dict(d=d, x=x, dy=dz, pro_y=y, pro_coef=coef, dw=torch.zeros(Co, 1, 1, Ci, device=DEV), **kw))
        dec.append(dict(d=d, x=x, dy=dz, dw=torch.zeros(Co, 1, 1, Ci, device=DEV), gram=torch.zeros(Ci * Ci + 1, device=DEV),
                        xsum=torch.zeros(Ci, device=DEV), wk=w, coef=coef, **kw))
        # float64: the staged operands exactly as the kernels see them (x after its prologue, rounded to T)
        xs = x.double()
        if pro:
            xs = torch.relu(torch.addcmul(sh.double(), xs, sc.double()).to(T).double()) if dtype == 1 else torch.relu(xs * sc.double() + sh.double())
        al, be, ga = coef.double().view(3, Co)
        dy = al * dz.double() + be * y.double() + ga
        refs.append(torch.einsum("nhwo,nhwi->oi", dy, xs))
    g2, gd = ops.wgrad_group_plan(dtype, two), ops.wgrad_group_plan(dtype, dec)
    assert gd.finish is not None and gd.njobs == 2 * len(dec) and g2.finish is None
    ops.wgrad_group_run(g2)
    for rep in range(2):                    # twice: the finish launch leaves gram / xsum zeroed for the next step
        for j in dec:
            j["dw"].zero_()
        ops.wgrad_group_run(gd)
        torch.cuda.synchronize()
        for (Ci, Co, H, N, pro), a, b, r in zip(shapes, two, dec, refs):
            assert float(b["gram"].abs().max()) == 0.0 and float(b["xsum"].abs().max()) == 0.0
            e2 = ((a["dw"].double().view(Co, Ci) - r).norm() / r.norm()).item()
            ed = ((b["dw"].double().view(Co, Ci) - r).norm() / r.norm()).item()
            print(f"{Ci}->{Co} H{H} N{N} pro={pro} {'bf16' if dtype else 'f32'}: rel err vs float64: two-tensor {e2:.2e}, decomposed {ed:.2e}")
            assert ed < (2e-3 if dtype == 1 else 2e-5), (Ci, Co, ed)
            assert ed < 2 * e2 + (1e-3 if dtype == 1 else 1e-6), (Ci, Co, ed, e2)


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
def test_block_merge_mask_bits(dtype):
    from frx import ops
    rows, Cc = 37 * 49, 256
    T = ops.TORCH_DT[dtype]
    y3 = _mk(dtype, rows, Cc, seed=1).to(DEV); idn = _mk(dtype, rows, Cc, seed=2).to(DEV)
    g = torch.Generator().manual_seed(3)
    s3, b3 = (torch.rand(Cc, generator=g) + 0.5).to(DEV), (torch.randn(Cc, generator=g) * 0.3).to(DEV)
    ref = torch.empty_like(y3); out = torch.empty_like(y3)
    V = 8 if dtype == 1 else 4
    mask = torch.zeros(rows * Cc // V, dtype=torch.uint8, device=DEV)
    ops.block_merge_fwd(dtype, rows, Cc, y3, s3, b3, idn, ref)
    ops.block_merge_fwd(dtype, rows, Cc, y3, s3, b3, idn, out, mask=mask)
    assert torch.equal(out, ref)
    want = ((ref.float() > 0).view(-1, V).to(torch.int32) << torch.arange(V, device=DEV, dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(mask, want)


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("geom", [(4, 56, 64), (3, 14, 64), (2, 9, 32)], ids=["56", "14", "odd9"])
def test_fused_stem_backward(dtype, geom):
    """frx_stem_bwd_reduce / frx_stem_bwd_apply (pool gather feeding the ReLU mask and the BatchNorm backward) == the
    stand-alone frx_stem_pool_bwd + frx_bn_bwd_reduce(relu) + frx_bn_bwd_apply(relu); in fp32 also == ATen's max_pool2d /
    ReLU backward followed by the closed-form BatchNorm backward."""
    from frx import ops
    N, H, Cc = geom
    Ho = (H + 2 - 3) // 2 + 1
    tdt = ops.TORCH_DT[dtype]
    g = torch.Generator().manual_seed(H)
    y = torch.randn(N, H, H, Cc, generator=g).to(tdt).to(DEV)
    scale = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cc, generator=g) * 0.3).to(DEV)
    mean = (torch.randn(Cc, generator=g) * 0.2).to(DEV)
    invstd = (torch.rand(Cc, generator=g) + 0.7).to(DEV)
    gamma = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    dout = torch.randn(N, Ho, Ho, Cc, generator=g).to(tdt).to(DEV)
    pooled = torch.empty(N, Ho, Ho, Cc, dtype=tdt, device=DEV)
    arg = torch.empty(N, Ho, Ho, Cc, dtype=torch.uint8, device=DEV)
    ops.stem_pool_fwd(dtype, N, H, H, Cc, y, scale, shift, pooled, arg)
    rows = N * H * H
    # stand-alone chain
    dpost = torch.empty_like(y)
    ops.stem_pool_bwd(dtype, N, H, H, Cc, dout, arg, dpost)
    nblk = ops.bn_bwd_partial_rows(rows, Cc)
    part = torch.zeros(max(nblk, ops.stem_bwd_partial_rows()) * 2 * Cc, device=DEV)
    ops.bn_bwd_reduce(dtype, rows, Cc, dpost, y, mean, invstd, part, scale=scale, shift=shift, relu=True)
    dga, dbe, coef = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * Cc, device=DEV)
    ops.bn_bwd_finalize(part, nblk, Cc, rows, gamma, mean, invstd, dga, dbe, coef)
    dy_ref = torch.empty_like(y)
    ops.bn_bwd_apply(dtype, rows, Cc, dpost, y, mean, invstd, coef, dy_ref, scale=scale, shift=shift, relu=True)
    # fused chain
    part2 = torch.zeros_like(part)
    ops.stem_bwd_reduce(dtype, N, H, H, Cc, dout, arg, y, scale, shift, mean, invstd, part2)
    dga2, dbe2, coef2 = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * Cc, device=DEV)
    ops.bn_bwd_finalize(part2, ops.stem_bwd_partial_rows(), Cc, rows, gamma, mean, invstd, dga2, dbe2, coef2)
    dy = torch.empty_like(y)
    ops.stem_bwd_apply(dtype, N, H, H, Cc, dout, arg, y, scale, shift, coef2, dy)
    torch.cuda.synchronize()
    for a, b, what in ((dga2, dga, "dgamma"), (dbe2, dbe, "dbeta"), (coef2, coef, "coef")):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-6, what
    # same dz, coefficients equal to ~1e-6: dy agrees to an ulp of the storage type
    _close(dy, dy_ref.float().cpu(), dtype, "fused dy vs stand-alone")
    assert (dy.float() - dy_ref.float()).abs().max().item() <= (2e-6 if dtype == ops.F32 else 2 ** -7) * dy_ref.float().abs().max().item()
    if dtype == ops.F32:
        yc = y.cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        sc, sh = scale.cpu().view(1, -1, 1, 1), shift.cpu().view(1, -1, 1, 1)
        p = F.max_pool2d(F.relu(yc * sc + sh), 3, 2, 1)
        (dpre,) = torch.autograd.grad(p, yc, dout.cpu().permute(0, 3, 1, 2))          # = dz * scale (chain through the affine)
        dz = dpre / sc
        xhat = (yc.detach() - mean.cpu().view(1, -1, 1, 1)) * invstd.cpu().view(1, -1, 1, 1)
        m1, m2 = dz.mean((0, 2, 3), keepdim=True), (dz * xhat).mean((0, 2, 3), keepdim=True)
        want = (gamma.cpu() * invstd.cpu()).view(1, -1, 1, 1) * (dz - m1 - xhat * m2)
        _close(dy, want.permute(0, 2, 3, 1), dtype, "fused dy vs ATen")
        assert (dbe2.cpu() - dz.sum((0, 2, 3))).abs().max().item() < 1e-3 and (dga2.cpu() - (dz * xhat).sum((0, 2, 3))).abs().max().item() < 1e-3


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "bf16"])
def test_bn_bwd_reduce_broadcasts_a_pooled_gradient(dtype):
    """frx_bn_bwd_reduce(g_pool_hw = HW) on the [N, C] gradient of an average pool == frx_avgpool_bwd followed by the plain
    reduce, bit for bit (same dz, same partial sums) -- without the broadcast tensor in memory."""
    from frx import ops
    N, HW, Cc = 6, 16, 256
    T = ops.TORCH_DT[dtype]
    dpool = _mk(dtype, N, Cc, seed=1).to(DEV)
    y = _mk(dtype, N * HW, Cc, seed=2).to(DEV)
    out = torch.relu(_mk(dtype, N * HW, Cc, seed=3)).to(DEV)
    g = torch.Generator().manual_seed(4)
    mean, invstd = (torch.randn(Cc, generator=g) * 0.2).to(DEV), (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    rows = N * HW
    nblk = ops.bn_bwd_partial_rows(rows, Cc)
    gfull = torch.empty(rows, Cc, dtype=T, device=DEV)
    ops.avgpool_bwd(dtype, N, HW, Cc, dpool, gfull)
    p_ref, dz_ref = torch.zeros(nblk, 2, Cc, device=DEV), torch.empty_like(gfull)
    ops.bn_bwd_reduce(dtype, rows, Cc, gfull, y, mean, invstd, p_ref, out=out, dz_out=dz_ref)
    p, dz = torch.zeros_like(p_ref), torch.empty_like(gfull)
    ops.bn_bwd_reduce(dtype, rows, Cc, dpool, y, mean, invstd, p, out=out, dz_out=dz, g_pool_hw=HW)
    torch.cuda.synchronize()
    assert torch.equal(dz, dz_ref) and torch.equal(p, p_ref)
    assert (dz_ref.float().abs().sum() > 0) and (dz_ref == 0).any()
