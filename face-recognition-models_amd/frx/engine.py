"""Execution plan for the ResNet-50 (torchvision v1.5 topology, fc -> 512) + margin-head
training / embedding path on one MI355X, built on the libfrx C ABI.

Everything is pre-allocated for a fixed per-GPU batch: parameters, gradients and SGD momentum
live in three flat fp32 buffers (one fused optimiser launch, one contiguous all-reduce target);
activations are NHWC in the compute dtype (bf16 = speed mode, fp32 = parity mode).
A step only enqueues kernels on the current stream -- no allocation, no host sync -- so the
whole step is hipGraph-capturable (torch.cuda.graph).

Reference: backbone = torchvision resnet50 + nn.Linear(2048, 512) (main_code/utils/backbones.py:16-18),
wrappers criterion.py:303-325 (and siblings), step model_utils.py:176-187, SGD :557.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import os
import struct
import torch

from . import ops
from .ops import BF16, F32

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
FEATURE_DIM = 512
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


@dataclass
class ConvSpec:
    name: str            # torchvision state-dict prefix, e.g. "layer1.0.conv1"
    bn: str              # matching BatchNorm prefix, e.g. "layer1.0.bn1"
    Ci: int
    Co: int
    k: int
    stride: int
    Hi: int
    stem: bool = False
    # filled by the engine
    desc: object = None
    desc_c: object = None    # stride-2 1x1 convs: the same conv as a stride-1 one on the coarse grid (compact dgrad)
    w_off: int = 0       # offset of the fp32 master weight (KRSC) in the flat buffers
    w_numel: int = 0
    g_off: int = 0       # gamma offset;  beta = g_off + Co
    bn_idx: int = 0      # index into the per-BN state table
    wk: torch.Tensor = None
    wt: torch.Tensor = None
    y: torch.Tensor = None
    stat_rows: int = 0
    # BatchNorm statistics as replicated totals (csrc/bn_tot.h): forward (sum y, sum y^2) and backward (sum dz, sum dz*xhat)
    R: int = 8
    tot_f_buf: torch.Tensor = None
    tot_b_buf: torch.Tensor = None
    tot_f: object = None     # ops.bn_tot(...) views for the consumers
    tot_b: object = None
    patch_dgrad: bool = False    # 3x3 / stride 1 on the patch-mode kernel: its input gradient evaluates the BN backward itself
    x_norm: torch.Tensor = None  # 3x3 / stride 1 on the patch-mode kernel (training): the forward keeps relu(bn(x)) for the weight gradient
    gram: torch.Tensor = None    # decomposed weight gradient (_gram_ok): x^T x [Ci*Ci] + the finish launch's counter
    xsum: torch.Tensor = None    #   and the column sums of x [Ci]

    @property
    def Ho(self):
        return (self.Hi + 2 * (self.k // 2) - self.k) // self.stride + 1


@dataclass
class BlockSpec:
    conv1: ConvSpec
    conv2: ConvSpec
    conv3: ConvSpec
    down: ConvSpec | None
    out: torch.Tensor = None
    # backward buffers that live until the optimiser step (the grouped weight-gradient launch reads them late)
    dz3: torch.Tensor = None     # masked gradient of the block output (feeds bn3, the identity and the projection)
    dy2: torch.Tensor = None     # gradient of conv2's raw output
    dyc: dict = None             # conv name -> copy of dy kept by its dgrad (where _keeps_dy says so)
    coefs: list = None           # BN-backward coefficients of bn3 / bn2 / bn1 / downsample
    mask: torch.Tensor = None    # out > 0 as one byte per 16-byte channel group (the backward's merge-ReLU mask)


def resnet50_specs(H=112):
    stem = ConvSpec("conv1", "bn1", 3, 64, 7, 2, H, stem=True)
    h = (stem.Ho + 2 - 3) // 2 + 1          # after max-pool
    blocks, inpl = [], 64
    for li, (n, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(n):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"layer{li}.{b}"
            c1 = ConvSpec(p + ".conv1", p + ".bn1", inpl, planes, 1, 1, h)
            c2 = ConvSpec(p + ".conv2", p + ".bn2", planes, planes, 3, stride, h)
            c3 = ConvSpec(p + ".conv3", p + ".bn3", planes, planes * 4, 1, 1, c2.Ho)
            ds = None
            if b == 0:
                ds = ConvSpec(p + ".downsample.0", p + ".downsample.1", inpl, planes * 4, 1, stride, h)
            blocks.append(BlockSpec(c1, c2, c3, ds))
            inpl, h = planes * 4, c2.Ho
    return stem, blocks, h


class ResNet50Engine:
    """Owns parameters, activations and scratch of the backbone; runs forward / backward."""

    def __init__(self, batch, dtype=BF16, device="cuda:0", H=112, extra_params=0, share=None):
        """share: another ResNet50Engine (same dtype / device) whose parameters, gradients, momentum,
        BN buffers and kernel-format weights this plan re-uses -- a plan per batch size, one model."""
        self.N, self.dtype, self.device, self.H = batch, dtype, torch.device(device), H
        self.tdt = ops.TORCH_DT[dtype]
        self.stem, self.blocks, self.h_final = resnet50_specs(H)
        self.convs = [self.stem] + [c for b in self.blocks for c in (b.conv1, b.conv2, b.conv3, b.down) if c is not None]
        # ---- flat parameter layout: [conv weights | bn gamma,beta | fc weight | fc bias | extra (head)]
        off = 0
        for c in self.convs:
            c.w_numel = c.Co * 7 * 8 * 4 if c.stem else c.Co * c.k * c.k * c.Ci
            c.w_off = off
            off += c.w_numel
        for i, c in enumerate(self.convs):
            c.g_off, c.bn_idx = off, i
            off += 2 * c.Co
        self.fc_w_off = off
        off += FEATURE_DIM * 2048
        self.fc_b_off = off
        off += FEATURE_DIM
        self.backbone_numel = off
        off = (off + 63) // 64 * 64
        self.extra_off = off
        self.n_params = (off + extra_params + 3) // 4 * 4
        dev = self.device
        if share is not None:
            assert share.dtype == dtype and share.device == self.device and share.n_params == self.n_params
            self.params, self.grads, self.mom = share.params, share.grads, share.mom
        else:
            self.params = torch.zeros(self.n_params, device=dev)
            self.grads = torch.zeros(self.n_params, device=dev)
            self.mom = torch.zeros(self.n_params, device=dev)
        # ---- BN buffers: running stats + batch statistics / affine of the current step
        ctot = sum(c.Co for c in self.convs)
        self.bn_off = {}
        o = 0
        for c in self.convs:
            self.bn_off[c.bn] = o
            o += c.Co
        if share is not None:
            self.running_mean, self.running_var = share.running_mean, share.running_var
            self.num_batches_tracked = share.num_batches_tracked
        else:
            self.running_mean = torch.zeros(ctot, device=dev)
            self.running_var = torch.ones(ctot, device=dev)
            self.num_batches_tracked = torch.zeros(len(self.convs), dtype=torch.int64, device=dev)
        self.bn_mean = torch.zeros(ctot, device=dev)
        self.bn_invstd = torch.zeros(ctot, device=dev)
        self.bn_scale = torch.zeros(ctot, device=dev)
        self.bn_shift = torch.zeros(ctot, device=dev)
        # ---- kernel-format weights, activations
        N = batch
        for ci_, c in enumerate(self.convs):
            if c.stem:
                c.desc = ops.conv_desc(dtype, N, H, H, 3, 64, 7, 7, 2, 3, stem=True)
            else:
                c.desc = ops.conv_desc(dtype, N, c.Hi, c.Hi, c.Ci, c.Co, c.k, c.k, c.stride, c.k // 2)
                # (the per-tile partial rows of the deterministic form are laid out by frx_conv_tile's row tile.  Measured per
                # layer inside a training step, fused dgrad vs bn_bwd_apply + prologue-free dgrad: 128- and 256-channel layers
                # at batch 256 35.0 vs 32.4 + 9 us, 38.2 vs 35.5 + 8; the 64-column tile (two blocks per CU next to three
                # patch buffers) 71.7 vs 43.2 + 20 and layer4's 64-pixel tile 53.8 vs 39.6 + 7 keep the separate pass)
                pm = ops.conv_patch_mode(c.desc, True) if (c.k == 3 and c.stride == 1) else 0
                sel = os.environ.get("FRX_PATCH_DGRAD", "1")      # 0: never; 2: wherever the kernel can (measurement)
                c.patch_dgrad = (pm != 0 and pm == ops._igemm_tile(c.desc, True)[0] and sel != "0"
                                 and (sel == "2" or (pm == 128 and c.Ci % 128 == 0)))
                if c.k == 1 and c.stride == 2:
                    c.desc_c = ops.conv_desc(dtype, N, c.Ho, c.Ho, c.Ci, c.Co, 1, 1, 1, 0)
            if share is not None:
                c.wk, c.wt = share.convs[ci_].wk, share.convs[ci_].wt
            elif c.stem:
                c.wk = torch.zeros(64, 7, 8, 4, dtype=self.tdt, device=dev)
            else:
                c.wk = torch.zeros(c.Co, c.k, c.k, c.Ci, dtype=self.tdt, device=dev)
                c.wt = torch.zeros(c.Ci, c.k, c.k, c.Co, dtype=self.tdt, device=dev)
            c.stat_rows = ops.conv_stat_rows(c.desc)
            c.y = torch.zeros(N, c.Ho, c.Ho, c.Co, dtype=self.tdt, device=dev)
        hp, wp = ops.stem_padded_dims(H, H)
        self.xin = torch.zeros(N, hp, wp, 4, dtype=self.tdt, device=dev)
        self.hpool = (self.stem.Ho + 2 - 3) // 2 + 1
        self.pool_out = torch.zeros(N, self.hpool, self.hpool, 64, dtype=self.tdt, device=dev)
        self.pool_arg = torch.zeros(N, self.hpool, self.hpool, 64, dtype=torch.uint8, device=dev)
        for b in self.blocks:
            b.out = torch.zeros_like(b.conv3.y)
            b.mask = torch.zeros(b.out.numel() // (8 if dtype == BF16 else 4), dtype=torch.uint8, device=dev)
            b.coefs = [torch.zeros(3 * c.Co, device=dev) if c is not None else None for c in (b.conv3, b.conv2, b.conv1, b.down)]
        self._train_ready = False       # the backward's per-block buffers: allocated by the first TRAINING forward (_ensure_training_buffers)
        self.pooled = torch.zeros(N, 2048, dtype=self.tdt, device=dev)
        self.feats = torch.zeros(N, FEATURE_DIM, device=dev)
        self.fc_desc = ops.conv_desc(dtype, N, 1, 1, 2048, FEATURE_DIM, 1, 1, 1, 0)
        if share is not None:
            self.fc_wk, self.fc_wt = share.fc_wk, share.fc_wt
        else:
            self.fc_wk = torch.zeros(FEATURE_DIM, 2048, dtype=self.tdt, device=dev)
            self.fc_wt = torch.zeros(2048, FEATURE_DIM, dtype=self.tdt, device=dev)
        # ---- scratch
        max_rows = max(c.stat_rows * c.Co for c in self.convs)
        self.stat_partial = torch.zeros(2 * max_rows, device=dev)
        max_act = max(c.y.numel() for c in self.convs)
        self.scratch = [torch.zeros(max_act, dtype=self.tdt, device=dev) for _ in range(5)]
        max_bp = max(ops.bn_bwd_partial_rows(c.y.numel() // c.Co, c.Co) * c.Co for c in self.convs)
        max_bp = max([max_bp, ops.stem_bwd_partial_rows() * 64] + [ops.conv_dgrad_stat_rows(c.desc) * c.Ci for c in self.convs if not c.stem])
        self.bwd_partial = torch.zeros(2 * max_bp, device=dev)
        self.coef = torch.zeros(3 * 2048, device=dev)
        self.g_pool = torch.zeros_like(self.pool_out)        # gradient w.r.t. the max-pool output
        self.dy_stem = torch.zeros_like(self.stem.y)
        self.grouped_wgrad = os.environ.get("FRX_WGRAD_GROUPED", "1") != "0"
        self.join_after_upper = True       # (kept in graph_key(): a data-parallel plan ends a segment after the upper list)
        # the projection branch of a layer's first block (conv + BN forward; BN-backward reduce + input / weight gradient
        # backward) is independent of the conv1-conv2-conv3 chain until the merge / until conv1's input gradient: it runs on a
        # side stream next to the chain (fork / join inside the captured step).  Replicated-totals mode only (the partial-rows
        # form shares one scratch buffer between the branches).  FRX_BRANCH_STREAM=0: in line.
        self.branch_stream = (torch.cuda.Stream(device=dev) if (dev.type == "cuda" and os.environ.get("FRX_BRANCH_STREAM", "1") != "0")
                              else None)
        # bf16 speed mode: BatchNorm statistics travel as replicated totals the producers add into with float atomics
        # and every consumer derives its constants from (no finalize launch per layer: -104 launches per step).  The fp32
        # parity mode and FRX_BN_DETERMINISTIC=1 keep partial rows + finalize launches (bit-reproducible sums).
        self.fused_bn = (dtype == BF16 and self.grouped_wgrad and os.environ.get("FRX_BN_DETERMINISTIC", "0") != "1")
        if self.fused_bn:
            self._plan_bn_totals()
        self.merge_fuse = os.environ.get("FRX_MERGE_FUSE", "1")
        self.fused_stem_bwd = True         # (False: max-pool backward + stand-alone BN backward, the form the fused stem kernels are tested against)
        self.mask_bits = True              # (False: the backward re-reads the block output instead of its 1-bit mask)
        self._wg_groups = None                               # planned at the end of __init__ (needs every buffer)
        self.dfeat_t = torch.zeros(N, FEATURE_DIM, dtype=self.tdt, device=dev)
        self.lr_dev = torch.zeros(1, device=dev)
        self.training = True
        self._eval_affine_ready = False
        self.share = share
        if share is None:
            self.reset_parameters()

    def _ensure_training_buffers(self):
        """Per-block buffers only a training step touches -- the masked output gradient, conv2's dy, the kept dy copies and
        the kept relu(bn(x)) input of each patch-mode conv2 (+300 MB at batch 256, linear in the batch) -- and the
        weight-gradient work lists that point at them.  Built by the first training forward (an eager call: the plan's
        host->device copy must not land inside a graph capture); embedding-only engines (the B = 512 engine of the LFW path,
        weight-sharing plans of other batch sizes that never train) never pay for them."""
        if self._train_ready:
            return
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise ops.FrxError("the first training forward of an engine allocates its backward buffers: run one eager step "
                               "before capturing a graph")
        for b in self.blocks:
            b.dz3 = torch.zeros_like(b.conv3.y)
            b.dy2 = torch.zeros_like(b.conv2.y)
            c2 = b.conv2
            pm = ops.conv_patch_mode(c2.desc, False) if (c2.k == 3 and c2.stride == 1) else 0
            if pm and pm == ops._igemm_tile(c2.desc)[0]:
                c2.x_norm = torch.zeros_like(b.conv1.y)      # (the grouped lists: -78 us per step)
            b.dyc = {c.name: torch.zeros_like(c.y) for c in (b.conv1, b.conv3, b.down) if c is not None and self._keeps_dy(c)}
            for c in (b.conv3, b.down):
                if c is not None and self._gram_ok(c):      # decomposed weight gradient: x^T x (+ the finish launch's counter), sum(x)
                    c.gram = torch.zeros(c.Ci * c.Ci + 1, device=self.device)
                    c.xsum = torch.zeros(c.Ci, device=self.device)
        self._train_ready = True
        if self.grouped_wgrad:
            self._plan_wgrad_groups()

    # ------------------------------------------------------------------ BatchNorm statistics as replicated totals
    def _plan_bn_totals(self):
        dev = self.device
        fed_by_reduce = {b.down.name for b in self.blocks if b.down is not None} | {self.blocks[-1].conv3.name}
        for c in self.convs:
            # Replicas: a producer block adds into row (its row-tile index mod R), and EVERY consumer block reads all R rows
            # of every channel it needs -- so few rows where the layer has few row tiles (layer3: 98, layer4: 32: their
            # adds are spread over the producer's run time anyway), 8 where it has hundreds, 16 for the stem (6272 tiles).
            # The layers whose backward sums come from a stand-alone reduce launch (512 blocks that finish together: the
            # projections' BatchNorms, the last bn3) keep 8.
            tiles = (c.y.numel() // c.Co + 127) // 128
            big = 8                                            # (4 measures the same, 2 and 16 worse: DESIGN.md 8.0)
            c.R = 2 * big if c.stem else (big if (tiles >= 256 or c.name in fed_by_reduce) else (2 if tiles >= 64 else 1))
        tot = sum(c.R * 2 * c.Co for c in self.convs)
        self.bn_tot_f = torch.zeros(tot, device=dev)
        self.bn_tot_b = torch.zeros(tot, device=dev)
        o = 0
        for c in self.convs:
            n = c.R * 2 * c.Co
            c.tot_f_buf, c.tot_b_buf = self.bn_tot_f[o:o + n], self.bn_tot_b[o:o + n]
            o += n
            count = c.y.numel() // c.Co
            c.tot_f = ops.bn_tot(c.tot_f_buf, c.R, count, self.gamma(c), beta=self.beta(c), eps=BN_EPS)
            c.tot_b = ops.bn_tot(c.tot_b_buf, c.R, count, self.gamma(c), mean=self._bn(self.bn_mean, c),
                                 invstd=self._bn(self.bn_invstd, c))
        f2i = lambda v: struct.unpack("i", struct.pack("f", v))[0]

        def table(convs, fwd):
            rows, blk = [], 0
            for c in convs:
                count = c.y.numel() // c.Co
                if fwd:
                    rows.append((c.tot_f_buf.data_ptr(), c.R, c.Co, count, self.gamma(c).data_ptr(), self.beta(c).data_ptr(),
                                 self._bn(self.running_mean, c).data_ptr(), self._bn(self.running_var, c).data_ptr(),
                                 self._bn(self.bn_mean, c).data_ptr(), self._bn(self.bn_invstd, c).data_ptr(),
                                 self._bn(self.bn_scale, c).data_ptr(), self._bn(self.bn_shift, c).data_ptr(), blk,
                                 f2i(BN_EPS), f2i(BN_MOMENTUM), self.num_batches_tracked[c.bn_idx:c.bn_idx + 1].data_ptr()))
                else:
                    rows.append((c.tot_b_buf.data_ptr(), c.R, c.Co, count, self.gamma(c).data_ptr(),
                                 self._bn(self.bn_mean, c).data_ptr(), self._bn(self.bn_invstd, c).data_ptr(),
                                 self.gamma(c, self.grads).data_ptr(), self.beta(c, self.grads).data_ptr(),
                                 self._coef_of[c.name].data_ptr(), 0, 0, blk, 0, 0, 0))
                blk += (c.Co + 255) // 256
            return torch.tensor(rows, dtype=torch.int64, device=dev), len(rows), blk
        # coefficient arrays: the blocks' own (the grouped weight gradients read them), the stem's in self.coef
        self._coef_of = {self.stem.name: self.coef}
        for b in self.blocks:
            for c, k in ((b.conv3, 0), (b.conv2, 1), (b.conv1, 2), (b.down, 3)):
                if c is not None:
                    self._coef_of[c.name] = b.coefs[k]
        self._bn_fin_fwd = table(self.convs, True)
        groups = [[], [], []]
        for bi, b in enumerate(self.blocks):
            g = groups[0] if bi >= self.SPLIT_BLOCK else (groups[1] if bi >= LAYERS[0] else groups[2])
            g += [c for c in (b.conv3, b.conv2, b.conv1, b.down) if c is not None]
        groups[2].append(self.stem)
        self._bn_fin_bwd = [table(g, False) for g in groups]

    # ------------------------------------------------------------------ parameter views
    def w_master(self, c: ConvSpec):
        shape = (c.Co, 7, 8, 4) if c.stem else (c.Co, c.k, c.k, c.Ci)
        return self.params[c.w_off:c.w_off + c.w_numel].view(shape)

    def w_grad(self, c: ConvSpec, buf=None):
        shape = (c.Co, 7, 8, 4) if c.stem else (c.Co, c.k, c.k, c.Ci)
        return (self.grads if buf is None else buf)[c.w_off:c.w_off + c.w_numel].view(shape)

    def gamma(self, c, buf=None):
        return (self.params if buf is None else buf)[c.g_off:c.g_off + c.Co]

    def beta(self, c, buf=None):
        return (self.params if buf is None else buf)[c.g_off + c.Co:c.g_off + 2 * c.Co]

    def fc_w(self, buf=None):
        return (self.params if buf is None else buf)[self.fc_w_off:self.fc_w_off + FEATURE_DIM * 2048].view(FEATURE_DIM, 2048)

    def fc_b(self, buf=None):
        return (self.params if buf is None else buf)[self.fc_b_off:self.fc_b_off + FEATURE_DIM]

    def extra(self, numel, buf=None):
        return (self.params if buf is None else buf)[self.extra_off:self.extra_off + numel]

    def _bn(self, t, c):
        o = self.bn_off[c.bn]
        return t[o:o + c.Co]

    def reset_parameters(self, seed=None):
        """torchvision init: conv kaiming_normal(fan_out, relu), BN gamma=1 beta=0, fc default Linear."""
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed)
        for c in self.convs:
            k = 7 if c.stem else c.k
            std = math.sqrt(2.0 / (c.Co * k * k))
            if c.stem:
                w = torch.zeros(64, 7, 8, 4)
                w[:, :, :7, :3] = torch.randn(64, 7, 7, 3, generator=g) * std
            else:
                w = torch.randn(c.Co, c.k, c.k, c.Ci, generator=g) * std
            self.w_master(c).copy_(w)
            self.gamma(c).fill_(1.0)
            self.beta(c).zero_()
        bound = 1.0 / math.sqrt(2048)
        self.fc_w().copy_((torch.rand(FEATURE_DIM, 2048, generator=g) * 2 - 1) * bound)
        self.fc_b().copy_((torch.rand(FEATURE_DIM, generator=g) * 2 - 1) * bound)
        self.running_mean.zero_()
        self.running_var.fill_(1.0)
        self.num_batches_tracked.zero_()
        self.mom.zero_()
        self.sync_weights()

    def _build_prep_table(self):
        rows, blk = [], 0
        for c in self.convs:
            if c.stem:      # [64][7][8][4] is copied as-is (mode 0: 1024 elements per block)
                rows.append((c.w_off, 64, 1, 224, c.wk.data_ptr(), 0, blk, 0))
                blk += (64 * 224 + 1023) // 1024
            else:           # mode 1: one 64x64 (co x ci) tile of one tap per block
                rows.append((c.w_off, c.Co, c.k * c.k, c.Ci, c.wk.data_ptr(), c.wt.data_ptr(), blk, 1))
                blk += (c.Co // 64) * (c.Ci // 64) * c.k * c.k
        rows.append((self.fc_w_off, FEATURE_DIM, 1, 2048, self.fc_wk.data_ptr(), self.fc_wt.data_ptr(), blk, 1))
        blk += (FEATURE_DIM // 64) * (2048 // 64)
        self._prep_rows, self._prep_blocks = len(rows), blk
        # the fused optimiser launch (frx_sgd_step_prep) walks the same table plus the ranges that have no kernel-format copy:
        # BatchNorm gamma / beta, then fc bias + padding + the margin head
        first_g = self.convs[0].g_off
        for lo, hi in ((first_g, self.fc_w_off), (self.fc_b_off, self.n_params)):
            assert lo % 4 == 0 and (hi - lo) % 4 == 0
            rows.append((lo, hi - lo, 0, 0, 0, 0, blk, 2))
            blk += (hi - lo + 1023) // 1024
        assert self.convs[-1].w_off + self.convs[-1].w_numel == first_g
        self._prep_table = torch.tensor(rows, dtype=torch.int64, device=self.device)
        self._sgd_blocks = blk

    def sync_weights(self, pad=True):
        """fp32 master -> kernel-format copies (after an optimiser step or a state-dict load): one launch.
        pad=False skips re-zeroing the stem's padding slots: the engine's own SGD step cannot move them (their
        gradient slots are never written, so weight decay and momentum act on exact zeros)."""
        if pad:
            m = self.w_master(self.stem)
            m[:, :, 7, :] = 0          # the stem's 8th tap / 4th channel exist only as padding
            m[..., 3] = 0
        if getattr(self, "_prep_table", None) is None:
            self._build_prep_table()
        ops.weight_prep_batched(self.dtype, self._prep_table[:self._prep_rows], self.params, self._prep_blocks)
        self._bump_weights_version()

    def _bump_weights_version(self):
        owner = self.share or self
        owner.weights_version = getattr(owner, "weights_version", 0) + 1

    # ------------------------------------------------------------------ forward
    def _conv_bn(self, c: ConvSpec, x, prev: ConvSpec | None):
        """y = conv(f_prev(x)); then this layer's BN statistics -> scale/shift."""
        kw = {}
        if prev is not None:
            kw = dict(in_scale=self._bn(self.bn_scale, prev), in_shift=self._bn(self.bn_shift, prev), in_relu=True)
        if self.training and c.x_norm is not None:
            # patch-mode 3x3: the prologue's output is kept (the weight gradient then stages it without a prologue per tap)
            if self.fused_bn:
                ops.conv_fwd_keep(c.desc, x, c.wk, c.y, c.x_norm, in_bn=prev.tot_f, stat_totals=c.tot_f_buf, stat_replicas=c.R)
                return c.y
            ops.conv_fwd_keep(c.desc, x, c.wk, c.y, c.x_norm, stat_partial=self.stat_partial, **kw)
            count = c.y.numel() // c.Co
            ops.bn_finalize(self.stat_partial, c.stat_rows, c.Co, count, self.gamma(c), self.beta(c),
                            self._bn(self.running_mean, c), self._bn(self.running_var, c),
                            self._bn(self.bn_mean, c), self._bn(self.bn_invstd, c),
                            self._bn(self.bn_scale, c), self._bn(self.bn_shift, c), BN_EPS, BN_MOMENTUM)
            return c.y
        if self.training and self.fused_bn:
            ops.conv_fwd_tot(c.desc, x, c.wk, c.y, in_bn=None if prev is None else prev.tot_f, stat_totals=c.tot_f_buf,
                             stat_replicas=c.R)
        elif self.training:
            ops.conv_fwd(c.desc, x, c.wk, c.y, stat_partial=self.stat_partial, **kw)
            count = c.y.numel() // c.Co
            ops.bn_finalize(self.stat_partial, c.stat_rows, c.Co, count, self.gamma(c), self.beta(c),
                            self._bn(self.running_mean, c), self._bn(self.running_var, c),
                            self._bn(self.bn_mean, c), self._bn(self.bn_invstd, c),
                            self._bn(self.bn_scale, c), self._bn(self.bn_shift, c), BN_EPS, BN_MOMENTUM)
        else:
            ops.conv_fwd(c.desc, x, c.wk, c.y, **kw)
        return c.y

    def _merge_fused(self, b):
        """whether block b's residual merge runs as the prologue of the next block's conv1 (replicated-totals training
        forward; FRX_MERGE_FUSE: "0" never, "all" wherever the next block has no projection, or the layers it pays in, e.g.
        "12" -- measured per layer inside a training step)"""
        sel = self.merge_fuse
        if sel == "0":
            return False
        layer = PLANES.index(b.conv1.Co) + 1
        return sel == "all" or str(layer) in sel

    def _merge_into_first(self, nxt):
        """a layer's first block: its conv1 takes the merge of the layer before as a prologue where the streamed kernel serves the
        shape (layer2's 256 -> 128: the largest merge of the network, 103 MB at batch 256); the block's projection then reads
        the stored block output behind it.  FRX_MERGE_FIRST=0: off"""
        c = nxt.conv1
        return (self.dtype == ops.BF16 and c.Ci == 256 and c.Co == 128 and os.environ.get("FRX_MERGE_FIRST", "1") != "0")

    def _prepare_eval_affine(self):
        for c in self.convs:
            ops.bn_eval_affine(self.gamma(c), self.beta(c), self._bn(self.running_mean, c),
                               self._bn(self.running_var, c), self._bn(self.bn_scale, c), self._bn(self.bn_shift, c), BN_EPS)

    def forward(self, images):
        """images: fp32 NCHW in [-1,1] or uint8 NHWC; returns feats [N,512] fp32 (engine-owned)."""
        N, dt = self.N, self.dtype
        if images.dim() != 4 or images.shape[0] != N:
            raise ops.FrxError(f"engine was planned for batch {N}, got an input of shape {tuple(images.shape)}")
        hw = tuple(images.shape[1:3]) if images.dtype == torch.uint8 else tuple(images.shape[2:4])
        if hw != (self.H, self.H):
            # (the reference's adaptive average pool takes any size; this plan's buffers are sized for one)
            raise ops.FrxError(f"engine was planned for {self.H}x{self.H} images, got {hw[0]}x{hw[1]}: resize in the "
                               f"transform (model_utils.py:539-547 feeds 112x112) or build an engine with H={hw[0]}")
        if not self.training:
            # batch statistics of a training step overwrite scale/shift, and running statistics move:
            # rebuild the eval affine whenever weights or statistics changed since it was made
            owner = self.share or self
            stamp = (getattr(owner, "weights_version", 0), int(getattr(owner, "stats_version", 0)))
            if self._eval_affine_ready != stamp:
                self._prepare_eval_affine()
                self._eval_affine_ready = stamp
        if self.training:
            self._ensure_training_buffers()
            if not self.fused_bn:       # (replicated totals: frx_bn_finalize_batched bumps the counters, no launch of its own)
                self.num_batches_tracked += 1
            owner = self.share or self
            owner.stats_version = getattr(owner, "stats_version", 0) + 1
            self._eval_affine_ready = False
        ops.input_prep(dt, images, self.xin)
        s = self.stem
        self._conv_bn(s, self.xin, None)
        fused = self.training and self.fused_bn
        if fused:
            ops.stem_pool_fwd_tot(dt, N, s.Ho, s.Ho, 64, s.y, s.tot_f, self.pool_out, self.pool_arg)
        else:
            ops.stem_pool_fwd(dt, N, s.Ho, s.Ho, 64, s.y, self._bn(self.bn_scale, s), self._bn(self.bn_shift, s),
                              self.pool_out, self.pool_arg)
        x = self.pool_out
        side = self.branch_stream if fused else None
        pending = None                  # (block, its input): a merge deferred into the next block's conv1 (frx_conv_fwd_merge)
        for bi, b in enumerate(self.blocks):
            proj_launched = False           # this block's projection (in line it waits until its output is needed)
            if side is not None and b.down is not None and pending is None:      # projection next to the chain
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    self._conv_bn(b.down, x, None)
                proj_launched = True
            if pending is not None:
                # the block before's out = relu(bn3(y3) + identity) is evaluated in this conv1's prologue and stored once on the
                # way (its merge pass: a read of the block output and a launch less per block)
                pb, px = pending
                pending = None
                if pb.down is not None and side is not None:
                    torch.cuda.current_stream(self.device).wait_stream(side)
                c1 = b.conv1
                ops.conv_fwd_merge(c1.desc, pb.conv3.y, pb.down.y if pb.down is not None else px, c1.wk, c1.y, pb.out,
                                   mask=pb.mask if self.mask_bits else None, bn3=pb.conv3.tot_f,
                                   bnd=pb.down.tot_f if pb.down is not None else None,
                                   stat_totals=c1.tot_f_buf, stat_replicas=c1.R)
                if b.down is not None:      # this block's projection reads the block output the launch above has just stored
                    if side is not None:
                        side.wait_stream(torch.cuda.current_stream(self.device))
                        with torch.cuda.stream(side):
                            self._conv_bn(b.down, x, None)
                    else:
                        self._conv_bn(b.down, x, None)
                    proj_launched = True
            else:
                self._conv_bn(b.conv1, x, None)
            self._conv_bn(b.conv2, b.conv1.y, b.conv1)
            self._conv_bn(b.conv3, b.conv2.y, b.conv2)
            rows = b.out.numel() // b.conv3.Co
            if fused:
                nxt = self.blocks[bi + 1] if bi + 1 < len(self.blocks) else None
                if nxt is not None and (nxt.down is None or self._merge_into_first(nxt)) and self._merge_fused(b):
                    if b.down is not None and not proj_launched:
                        self._conv_bn(b.down, x, None)
                    pending = (b, x)          # nothing reads b.out before the next conv1 has stored it
                    x = b.out
                    continue
                if b.down is not None:
                    if not proj_launched:
                        self._conv_bn(b.down, x, None)
                    elif side is not None:
                        torch.cuda.current_stream(self.device).wait_stream(side)
                ops.block_merge_fwd_tot(dt, rows, b.conv3.Co, b.conv3.y, b.conv3.tot_f, b.down.y if b.down is not None else x,
                                        b.out, bnd=b.down.tot_f if b.down is not None else None,
                                        mask=b.mask if self.mask_bits else None)
            elif b.down is not None:
                self._conv_bn(b.down, x, None)
                ops.block_merge_fwd(dt, rows, b.conv3.Co, b.conv3.y, self._bn(self.bn_scale, b.conv3),
                                    self._bn(self.bn_shift, b.conv3), b.down.y, b.out,
                                    sd=self._bn(self.bn_scale, b.down), bd=self._bn(self.bn_shift, b.down),
                                    mask=b.mask if (self.training and self.mask_bits) else None)
            else:
                ops.block_merge_fwd(dt, rows, b.conv3.Co, b.conv3.y, self._bn(self.bn_scale, b.conv3),
                                    self._bn(self.bn_shift, b.conv3), x, b.out, mask=b.mask if (self.training and self.mask_bits) else None)
            x = b.out
        if fused:       # one launch: the canonical mean / invstd / scale / shift arrays (the backward reads them), the
            ops.bn_finalize_batched(*self._bn_fin_fwd)      # running statistics, and the totals zeroed for the next step
        hw = self.h_final * self.h_final
        ops.avgpool_fwd(dt, N, hw, 2048, x, self.pooled)
        ops.conv_fwd(self.fc_desc, self.pooled, self.fc_wk, self.feats, bias=self.fc_b(), out_f32=True)
        return self.feats

    # ------------------------------------------------------------------ backward
    def _bn_backward(self, c: ConvSpec, g, dy, out=None, relu=False, dz_out=None):
        """g: upstream gradient w.r.t. the post-BN(-ReLU) tensor; writes dy (w.r.t. the conv output)."""
        rows = c.y.numel() // c.Co
        nblk = ops.bn_bwd_partial_rows(rows, c.Co)
        sc, sh = (self._bn(self.bn_scale, c), self._bn(self.bn_shift, c)) if relu else (None, None)
        mean, invstd = self._bn(self.bn_mean, c), self._bn(self.bn_invstd, c)
        ops.bn_bwd_reduce(self.dtype, rows, c.Co, g, c.y, mean, invstd, self.bwd_partial, out=out, scale=sc, shift=sh,
                          relu=relu, dz_out=dz_out)
        ops.bn_bwd_finalize(self.bwd_partial, nblk, c.Co, rows, self.gamma(c), mean, invstd, self.gamma(c, self.grads),
                            self.beta(c, self.grads), self.coef)
        src = dz_out if dz_out is not None else g
        if dz_out is not None:
            ops.bn_bwd_apply(self.dtype, rows, c.Co, src, c.y, mean, invstd, self.coef, dy)
        else:
            ops.bn_bwd_apply(self.dtype, rows, c.Co, src, c.y, mean, invstd, self.coef, dy, out=out, scale=sc, shift=sh, relu=relu)
        return dy

    def _like(self, buf, ref):
        return buf[:ref.numel()].view(ref.shape)

    def _finalize_bwd(self, c: ConvSpec, nrows_partial, coef):
        """partial sums (sum dz, sum dz*xhat) -> dgamma/dbeta (+=) and the affine coefficients of dy"""
        if self.fused_bn:       # the consumers derive them from c.tot_b; the batched launch before the weight gradients closes
            return
        rows = c.y.numel() // c.Co
        ops.bn_bwd_finalize(self.bwd_partial, nrows_partial, c.Co, rows, self.gamma(c), self._bn(self.bn_mean, c),
                            self._bn(self.bn_invstd, c), self.gamma(c, self.grads), self.beta(c, self.grads), coef)

    def _epi(self, c: ConvSpec, out=None):
        """keyword arguments of conv_dgrad_bn's epilogue for back-propagating through BN `c` (+ ReLU / merge)"""
        kw = dict(epi_y=c.y, epi_mean=self._bn(self.bn_mean, c), epi_invstd=self._bn(self.bn_invstd, c))
        if self.fused_bn:
            kw.update(epi_totals=c.tot_b_buf, epi_replicas=c.R)
        else:
            kw["epi_partial"] = self.bwd_partial
        if out is not None and out.dtype == torch.uint8:
            kw["epi_out_bits"] = out       # the block's 1-bit-per-element mask: 1/16 of reading its bf16 output again
        elif out is not None:
            kw["epi_out"] = out
        else:
            kw["epi_scale"], kw["epi_shift"] = self._bn(self.bn_scale, c), self._bn(self.bn_shift, c)
        return kw

    SPLIT_BLOCK = LAYERS[0] + LAYERS[1]     # first block of layer3: layers 3+4 hold 94 % of the conv parameters

    def backward(self, dfeat):
        self.backward_upper(dfeat)
        self.backward_lower()

    def grad_ranges(self):
        """flat-buffer ranges whose gradients are final after backward_upper() / backward_lower():
        data-parallel training all-reduces the (large) upper ranges while the lower backward still runs"""
        cut_w = self.blocks[self.SPLIT_BLOCK].conv1.w_off
        cut_g = self.blocks[self.SPLIT_BLOCK].conv1.g_off
        first_g = self.convs[0].g_off
        if getattr(self, "head_bucket", False):
            # the fc layer and the head (26 MB of the 120 at 10 575 classes) are final as soon as the head's backward and
            # the fc layer's are done: their all-reduce leaves before layer4's backward starts
            return {"head": [(self.fc_w_off, self.n_params)], "upper": [(cut_w, first_g), (cut_g, self.fc_w_off)],
                    "lower": [(0, cut_w), (first_g, cut_g)]}
        return {"upper": [(cut_w, first_g), (cut_g, self.n_params)], "lower": [(0, cut_w), (first_g, cut_g)]}

    def backward_upper(self, dfeat):
        """dfeat [N,512] fp32: gradient of the loss w.r.t. feats.  Accumulates into self.grads.

        BatchNorm backward is fused into the convolution backward wherever the conv is 1x1: the dgrad that
        produces a gradient masks it and reduces sum(dz), sum(dz*xhat) in its epilogue, and the consumers
        (dgrad / wgrad of the layer below) read dy = alpha*dz + beta*y + gam on the fly.  Only the 3x3 convs
        (whose 9 taps would re-evaluate the prologue 9x), the downsample BNs and the stem keep the stand-alone
        reduce / apply kernels."""
        self.backward_fc(dfeat)
        self.backward_upper_rest()

    def backward_fc(self, dfeat):
        """the fc layer's backward: its weight / bias gradients are final afterwards (with `head_bucket` its weight gradient
        is a launch of its own instead of a job of the upper grouped list, so that the bucket can leave early)"""
        dt, S = self.dtype, self.scratch
        (self.share or self)._grads_clean = False
        ops.cast(dt, dfeat, self.dfeat_t, to_f32=False)
        if not self.grouped_wgrad or getattr(self, "head_bucket", False):   # (grouped: one more job of the upper list --
            ops.conv_wgrad(self.fc_desc, self.pooled, self.dfeat_t, self.fc_w(self.grads))     # its own launch costs 34 us for 64 tiles)
        ops.colsum_f32(dfeat, self.fc_b(self.grads))
        ops.conv_dgrad(self.fc_desc, self.dfeat_t, self.fc_wt, self._like(S[4], self.pooled))

    def backward_upper_rest(self):
        N, dt = self.N, self.dtype
        S = self.scratch
        dpool = self._like(S[4], self.pooled)
        last = self.blocks[-1]
        # last block: its output gradient comes from the average pool, so mask + reduce run stand-alone; the reduce
        # broadcasts the pooled gradient itself (no [N, 4, 4, 2048] tensor, no launch for it)
        c3 = last.conv3
        rows3 = c3.y.numel() // c3.Co
        if self.fused_bn:
            ops.bn_bwd_reduce_tot(dt, rows3, c3.Co, dpool, c3.y, self._bn(self.bn_mean, c3), self._bn(self.bn_invstd, c3),
                                  c3.tot_b_buf, c3.R, out=last.out, dz_out=last.dz3, g_pool_hw=self.h_final * self.h_final)
        else:
            ops.bn_bwd_reduce(dt, rows3, c3.Co, dpool, c3.y, self._bn(self.bn_mean, c3), self._bn(self.bn_invstd, c3),
                              self.bwd_partial, out=last.out, dz_out=last.dz3, g_pool_hw=self.h_final * self.h_final)
        self._bw_npart = ops.bn_bwd_partial_rows(rows3, c3.Co)
        self._backward_blocks(len(self.blocks) - 1, self.SPLIT_BLOCK)
        self._close_bn_bwd(0)
        self._run_wgrad_group(0)

    def backward_lower(self):
        # (the weight-gradient lists of layer2 and of layer1 + stem run after ALL their input gradients: overlapping a list
        # with the input gradients below it on a side stream measured 0.03 ms slower, profiles/r03_wgrad_side_stream.txt)
        self._backward_blocks(self.SPLIT_BLOCK - 1, 0)
        self._backward_stem()
        self._close_bn_bwd(1)
        self._run_wgrad_group(1)
        self._close_bn_bwd(2)
        self._run_wgrad_group(2)

    def _close_bn_bwd(self, which):
        """replicated totals of one group of BatchNorm layers -> dgamma / dbeta (+=) and the coefficient arrays the grouped
        weight gradients read; the rows are zeroed for the next step (one launch)"""
        if self.fused_bn:
            ops.bn_bwd_finalize_batched(*self._bn_fin_bwd[which])

    def _backward_blocks(self, hi, lo):
        N, dt = self.N, self.dtype
        S = self.scratch
        npart = self._bw_npart
        for bi in range(hi, lo - 1, -1):
            b = self.blocks[bi]
            prev = self.blocks[bi - 1] if bi > 0 else None
            x_in = prev.out if prev is not None else self.pool_out
            c1, c2, c3, ds = b.conv1, b.conv2, b.conv3, b.down
            C3, C2, C1, CD = b.coefs
            dz3 = b.dz3
            side = self.branch_stream if (self.fused_bn and ds is not None) else None
            if side is not None:      # the projection's backward next to the conv3 / conv2 input gradients
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    proj = self._backward_projection(b, x_in, dz3, CD)
            self._finalize_bwd(c3, npart, C3)
            # conv3 (1x1): the dgrad evaluates dy3 = affine(dz3, y3) while staging its tiles and handles bn2 in its epilogue
            dz2 = self._like(S[4], c2.y)
            self._bwd_1x1(b, c3, dz3, C3, c2.y, dz2, x_bn=c2, **self._epi(c2))
            self._finalize_bwd(c2, ops.conv_dgrad_stat_rows(c3.desc), C2)
            # conv2 (3x3).  Stride 1 (patch-mode kernel: every dy2 element is transformed once per staged patch): the dgrad
            # evaluates dy2 = affine(dz2, y2) itself and stores it for the weight gradient on the way.  Stride 2 (nine taps
            # would re-evaluate a prologue nine times): dy2 is materialised by a pass of its own first.
            rows2 = c2.y.numel() // c2.Co
            dz1 = self._like(S[3] if c2.patch_dgrad else S[4], c1.y)      # (the fused form reads dz2 = S[4] while it writes)
            if c2.patch_dgrad:
                pro = dict(pro_tot=c2.tot_b) if self.fused_bn else dict(pro_coef=C2)
                ops.conv_dgrad_bn(c2.desc, dz2, c2.wt, dz1, pro_y=c2.y, pro_dy_out=b.dy2, **pro, **self._epi(c1))
                self._wgrad_conv2(b)
            else:
                if self.fused_bn:
                    ops.bn_bwd_apply_tot(dt, rows2, c2.Co, dz2, c2.y, c2.tot_b, b.dy2)
                else:
                    ops.bn_bwd_apply(dt, rows2, c2.Co, dz2, c2.y, self._bn(self.bn_mean, c2), self._bn(self.bn_invstd, c2), C2, b.dy2)
                self._wgrad_conv2(b)
                ops.conv_dgrad_bn(c2.desc, b.dy2, c2.wt, dz1, **self._epi(c1))
            self._finalize_bwd(c1, ops.conv_dgrad_stat_rows(c2.desc), C1)
            addend, add_stride = dz3, 0
            if ds is not None:
                if side is not None:
                    addend, add_stride = proj
                    torch.cuda.current_stream(self.device).wait_stream(side)
                else:
                    addend, add_stride = self._backward_projection(b, x_in, dz3, CD)
            # conv1 (1x1): its dgrad writes the masked output gradient of the block below (or the pool's)
            if prev is not None:      # epilogue: merge-ReLU mask of the block below + its bn3 reduce
                self._bwd_1x1(b, c1, dz1, C1, x_in, prev.dz3, addend=addend, addend_stride=add_stride,
                              **self._epi(prev.conv3, out=prev.mask if self.mask_bits else prev.out))
                npart = ops.conv_dgrad_stat_rows(c1.desc)
            else:
                self._bwd_1x1(b, c1, dz1, C1, x_in, self.g_pool, addend=addend, addend_stride=add_stride)
        self._bw_npart = npart

    def _backward_projection(self, b, x_in, dz3, CD):
        """BN-backward reduce of the projection's BatchNorm, then its input gradient (into scratch: conv1's dgrad adds it) and
        weight gradient; returns (addend, addend_stride)"""
        N, dt, S, ds = self.N, self.dtype, self.scratch, b.down
        rowsd = ds.y.numel() // ds.Co
        if self.fused_bn:
            ops.bn_bwd_reduce_tot(dt, rowsd, ds.Co, dz3, ds.y, self._bn(self.bn_mean, ds), self._bn(self.bn_invstd, ds),
                                  ds.tot_b_buf, ds.R)
        else:
            ops.bn_bwd_reduce(dt, rowsd, ds.Co, dz3, ds.y, self._bn(self.bn_mean, ds), self._bn(self.bn_invstd, ds),
                              self.bwd_partial)
        self._finalize_bwd(ds, ops.bn_bwd_partial_rows(rowsd, ds.Co), CD)
        if ds.desc_c is not None:
            # stride-2 projection: only the even pixels of its input gradient are non-zero.  Compute those as a
            # stride-1 conv on the coarse grid ([N,Ho,Wo,Ci], a quarter of the rows) and let conv1's dgrad add
            # them in place, instead of a full-size GEMM whose gather is 3/4 zeros.
            addend, add_stride = S[2][:N * ds.Ho * ds.Ho * ds.Ci].view(N, ds.Ho, ds.Ho, ds.Ci), 2
        else:
            addend, add_stride = self._like(S[2], x_in), 0
        self._bwd_1x1(b, ds, dz3, CD, x_in, addend)
        return addend, add_stride

    @staticmethod
    def _keeps_dy(c):
        """1x1 convs whose dgrad keeps a copy of dy = alpha*dz + beta*y + gam for the weight gradient: where writing dy
        once is cheaper than evaluating it twice -- narrow dy, or deep layers whose wgrad is ALU-bound (measured per
        layer type, scripts/layer_times.py).  Elsewhere (layer1/2 conv3, the projections: HBM-bound) both evaluate it."""
        return c.k == 1 and (c.Co <= c.Ci or (c.Co >= 1024 and c.stride == 1))

    def _bwd_1x1(self, b, c, dz, coef, x, dx, x_bn=None, addend=None, **epi):
        """input and weight gradient of a 1x1 conv whose BN backward is fused (dy = alpha*dz + beta*y + gam)"""
        dd = c.desc_c if c.desc_c is not None else c.desc      # dgrad geometry (compact for stride-2 projections)
        pro = dict(pro_tot=c.tot_b) if self.fused_bn else dict(pro_coef=coef)
        if self._keeps_dy(c):
            dy = b.dyc[c.name]
            ops.conv_dgrad_bn(dd, dz, c.wt, dx, addend=addend, pro_y=c.y, pro_dy_out=dy, **pro, **epi)
            self._wgrad(c, x, dy, x_bn=x_bn)
        else:
            self._wgrad(c, x, dz, x_bn=x_bn, pro_y=c.y, pro_coef=coef)
            ops.conv_dgrad_bn(dd, dz, c.wt, dx, addend=addend, pro_y=c.y, **pro, **epi)

    def _backward_stem(self):
        # stem: max-pool -> ReLU/BN -> conv weight gradient (no image gradient)
        N, dt = self.N, self.dtype
        S = self.scratch
        s = self.stem
        if self.fused_stem_bwd:
            # two passes that re-gather from the pooled gradient instead of storing the 56x56 gradient and reading it twice
            sc, sh = self._bn(self.bn_scale, s), self._bn(self.bn_shift, s)
            mean, invstd = self._bn(self.bn_mean, s), self._bn(self.bn_invstd, s)
            if self.fused_bn:
                ops.stem_bwd_reduce_tot(dt, N, s.Ho, s.Ho, 64, self.g_pool, self.pool_arg, s.y, sc, sh, mean, invstd, s.tot_b_buf, s.R)
                ops.stem_bwd_apply_tot(dt, N, s.Ho, s.Ho, 64, self.g_pool, self.pool_arg, s.y, sc, sh, s.tot_b, self.dy_stem)
            else:
                ops.stem_bwd_reduce(dt, N, s.Ho, s.Ho, 64, self.g_pool, self.pool_arg, s.y, sc, sh, mean, invstd, self.bwd_partial)
                ops.bn_bwd_finalize(self.bwd_partial, ops.stem_bwd_partial_rows(), 64, s.y.numel() // 64, self.gamma(s), mean, invstd,
                                    self.gamma(s, self.grads), self.beta(s, self.grads), self.coef)
                ops.stem_bwd_apply(dt, N, s.Ho, s.Ho, 64, self.g_pool, self.pool_arg, s.y, sc, sh, self.coef, self.dy_stem)
        else:
            dpost = self._like(S[2], s.y)
            ops.stem_pool_bwd(dt, N, s.Ho, s.Ho, 64, self.g_pool, self.pool_arg, dpost)
            self._bn_backward(s, dpost, self.dy_stem, relu=True)
        self._wgrad(s, self.xin, self.dy_stem)     # (padding tap / channel slots are not written)

    # ------------------------------------------------------------------ weight gradients
    def _gram_ok(self, c):
        """1x1 / stride-1 convs whose weight gradient runs DECOMPOSED in the grouped lists: the ones that would otherwise read
        dz AND the raw output y (two full-width tensors) -- dy = alpha*dz + beta*y + gam with y = x W^T linear in x gives
        dW = alpha (.) dz^T x + beta (.) W (x^T x) + gam (x) sum(x): the y read (103 MB per layer1 conv3 at batch 256)
        becomes a Ci x Ci side product of the x the list streams anyway.  Layer1's conv3 and projection (Ci = 64): same
        box, alternating, 6.246 / 6.246 / 6.246 -> 6.211 / 6.178 / 6.176 ms per step, the layer1 + stem list 376 -> 344 us.
        Layer2's conv3 (Ci = 128: a 128 x 128 side tile per pixel split) LOSES what it saves (that list 281 -> 322 us):
        not decomposed (profiles/r04_wgrad_gram_ab.txt).  FRX_WGRAD_GRAM (0 / 64 / 128): the measurement's switch."""
        lim = int(os.environ.get("FRX_WGRAD_GRAM", "64"))
        return self.grouped_wgrad and c.k == 1 and c.stride == 1 and not self._keeps_dy(c) and c.Ci <= min(lim, 128)

    def _wgrad_job(self, c, x, dy, x_bn=None, pro_y=None, pro_coef=None):
        pro = {} if x_bn is None else dict(in_scale=self._bn(self.bn_scale, x_bn), in_shift=self._bn(self.bn_shift, x_bn),
                                           in_relu=True)
        return dict(d=c.desc, x=x, dy=dy, dw=self.w_grad(c), pro_y=pro_y, pro_coef=pro_coef, **pro)

    def _wgrad_conv2(self, b):
        if b.conv2.x_norm is not None:
            self._wgrad(b.conv2, b.conv2.x_norm, b.dy2)
        else:
            self._wgrad(b.conv2, b.conv1.y, b.dy2, x_bn=b.conv1)

    def _wgrad(self, c, x, dy, x_bn=None, pro_y=None, pro_coef=None):
        """per-layer launch -- or nothing when the grouped launch covers this layer (same operands, planned once)"""
        if self.grouped_wgrad:
            return
        j = self._wgrad_job(c, x, dy, x_bn, pro_y, pro_coef)
        kw = {k: j[k] for k in ("in_scale", "in_shift", "in_relu") if k in j}
        if pro_y is not None:
            ops.conv_wgrad_bn(c.desc, x, dy, pro_y, pro_coef, j["dw"], **kw)
        else:
            ops.conv_wgrad(c.desc, x, dy, j["dw"], **kw)

    def _plan_wgrad_groups(self):
        """Three work lists of persistent-block weight-gradient items: [upper blocks] (data-parallel training all-reduces
        the upper gradients while the lower backward still runs), [layer2] and [layer1 + stem] -- the last one holds only
        64 x 64-tile layers, which frx_wgrad_group_run then runs four blocks per CU instead of two."""
        groups = [[], [], []]
        for bi, b in enumerate(self.blocks):
            jobs = groups[0] if bi >= self.SPLIT_BLOCK else (groups[1] if bi >= LAYERS[0] else groups[2])
            x_in = self.blocks[bi - 1].out if bi > 0 else self.pool_out
            C3, C2, C1, CD = b.coefs
            for c, x, xb, dz, coef in ((b.conv3, b.conv2.y, b.conv2, b.dz3, C3), (b.down, x_in, None, b.dz3, CD),
                                       (b.conv1, x_in, None, None, C1)):
                if c is None:
                    continue
                if self._keeps_dy(c):
                    jobs.append(self._wgrad_job(c, x, b.dyc[c.name], x_bn=xb))
                elif self._gram_ok(c):
                    jobs.append(dict(self._wgrad_job(c, x, dz, x_bn=xb), gram=c.gram, xsum=c.xsum, wk=c.wk, coef=coef))
                else:
                    jobs.append(self._wgrad_job(c, x, dz, x_bn=xb, pro_y=c.y, pro_coef=coef))
            if b.conv2.x_norm is not None:
                jobs.append(self._wgrad_job(b.conv2, b.conv2.x_norm, b.dy2))
            else:
                jobs.append(self._wgrad_job(b.conv2, b.conv1.y, b.dy2, x_bn=b.conv1))
        groups[2].append(self._wgrad_job(self.stem, self.xin, self.dy_stem))
        if not getattr(self, "head_bucket", False):
            groups[0].append(dict(d=self.fc_desc, x=self.pooled, dy=self.dfeat_t, dw=self.fc_w(self.grads), pro_y=None, pro_coef=None))
        self._wg_groups = [ops.wgrad_group_plan(self.dtype, jobs) for jobs in groups]

    def set_head_bucket(self, on):
        """data parallel: make the fc layer's gradients final right after the head's backward (grad_ranges()["head"])"""
        on = bool(on)
        if on != getattr(self, "head_bucket", False):
            self.head_bucket = on
            if self.grouped_wgrad and self._train_ready:
                torch.cuda.synchronize(self.device)
                self._old_wg_groups = self._wg_groups      # (a graph captured under the other setting may still name these tables)
                self._plan_wgrad_groups()

    def _run_wgrad_group(self, which):
        if not self.grouped_wgrad:
            return
        if self._wg_groups is None:
            self._plan_wgrad_groups()
        ops.wgrad_group_run(self._wg_groups[which])

    # ------------------------------------------------------------------ optimiser
    def zero_grad(self):
        self.grads.zero_()
        (self.share or self)._grads_clean = True

    def ensure_zero_grad(self):
        """optimizer.zero_grad() of model_utils.py:184 for the fused step: a launch only when the gradient buffer is not
        already zero (sgd_step() zeroes it as it consumes it)"""
        if not getattr(self.share or self, "_grads_clean", False):
            self.zero_grad()
        (self.share or self)._grads_clean = False            # the backward that follows accumulates into it

    def sgd_step(self, lr=None, momentum=0.9, weight_decay=5e-4, grad_scale=1.0, zero_grads=True):
        """lr=None: read the learning rate from self.lr_dev (graph-replay friendly).  ONE launch: SGD on the flat fp32
        buffers, the kernel-format (KRSC / CRSK) copies of the updated conv / fc weights, and -- zero_grads -- the zero-fill
        of the gradient buffer the next step accumulates into (model_utils.py:184-187)."""
        if getattr(self, "_prep_table", None) is None:
            self._build_prep_table()
        ops.sgd_step_prep(self.dtype, self._prep_table, self._sgd_blocks, self.params, self.grads, self.mom,
                          0.0 if lr is None else lr, momentum, weight_decay, grad_scale,
                          lr_dev=self.lr_dev if lr is None else None, zero_grads=zero_grads)
        (self.share or self)._grads_clean = bool(zero_grads)
        self._bump_weights_version()

    # ------------------------------------------------------------------ torchvision-compatible state dict
    def state_dict(self, prefix=""):
        sd = {}
        for i, c in enumerate(self.convs):
            w = self.w_master(c)
            w = w[:, :, :7, :3] if c.stem else w
            sd[prefix + c.name + ".weight"] = w.permute(0, 3, 1, 2).contiguous().clone()
            sd[prefix + c.bn + ".weight"] = self.gamma(c).clone()
            sd[prefix + c.bn + ".bias"] = self.beta(c).clone()
            sd[prefix + c.bn + ".running_mean"] = self._bn(self.running_mean, c).clone()
            sd[prefix + c.bn + ".running_var"] = self._bn(self.running_var, c).clone()
            sd[prefix + c.bn + ".num_batches_tracked"] = self.num_batches_tracked[i].clone()
        sd[prefix + "fc.weight"] = self.fc_w().clone()
        sd[prefix + "fc.bias"] = self.fc_b().clone()
        return sd

    def load_state_dict(self, sd, prefix="", strict=True):
        missing = []

        def get(k):
            if prefix + k not in sd:
                missing.append(prefix + k)
                return None
            return sd[prefix + k].to(self.device)
        for i, c in enumerate(self.convs):
            w = get(c.name + ".weight")
            if w is not None:
                w = w.float().permute(0, 2, 3, 1)
                if c.stem:
                    m = self.w_master(c)
                    m.zero_()
                    m[:, :, :7, :3] = w
                else:
                    self.w_master(c).copy_(w)
            for key, dst in ((".weight", self.gamma(c)), (".bias", self.beta(c)),
                             (".running_mean", self._bn(self.running_mean, c)),
                             (".running_var", self._bn(self.running_var, c))):
                v = get(c.bn + key)
                if v is not None:
                    dst.copy_(v.float())
            v = get(c.bn + ".num_batches_tracked")
            if v is not None:
                self.num_batches_tracked[i] = v
        for key, dst in (("fc.weight", self.fc_w()), ("fc.bias", self.fc_b())):
            v = get(key)
            if v is not None:
                dst.copy_(v.float())
        if strict and missing:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self.sync_weights()
        return missing


HEAD_KINDS = {"arcface": ops.ARC, "cosface": ops.COS, "sphereface": ops.SPHERE, "curricular": ops.CURR,
              "mv_am": ops.MV_AM, "mv_arc": ops.MV_ARC, "adaface": ops.ADA, "elastic_arc": ops.ELASTIC_ARC,
              "elastic_cos": ops.ELASTIC_COS, "magface": ops.MAG, "vpl_arcface": ops.VPL}
# (s, m) per head: main_code/utils/config.py:16-70.  SphereFace ignores s (criterion.py:119-123); the elastic heads
# sample their margin around m; MagFace derives it from the feature norm.
HEAD_DEFAULTS = {ops.ARC: (64.0, 0.5), ops.COS: (64.0, 0.35), ops.SPHERE: (1.0, 2.0), ops.CURR: (64.0, 0.5),
                 ops.MV_AM: (32.0, 0.35), ops.MV_ARC: (32.0, 0.35), ops.ADA: (64.0, 0.4), ops.ELASTIC_ARC: (64.0, 0.5),
                 ops.ELASTIC_COS: (64.0, 0.35), ops.MAG: (64.0, 0.0), ops.VPL: (64.0, 0.5)}
# frx_head_desc::p defaults: MV mv_weight (config.py:30); ADA h, t_alpha (:49-50); MAG l_margin, u_margin, l_a, u_a (:67-70)
# VPL lamda, delta (:43-44)
HEAD_P_DEFAULTS = {ops.MV_AM: (1.12,), ops.MV_ARC: (1.12,), ops.ADA: (0.333, 0.99), ops.MAG: (0.45, 0.8, 10.0, 110.0),
                   ops.VPL: (0.15, 100.0)}
HEAD_FLAG_DEFAULTS = {ops.VPL: 2}           # VPL: memory in use (norm_training_flag), easy_margin off (config.py:42)
ELASTIC_KINDS = (ops.ELASTIC_ARC, ops.ELASTIC_COS)


class FaceEngine:
    """Backbone + margin head + fused SGD: one training step = forward, CE, backward, update."""

    def __init__(self, kind, num_classes, batch, dtype=BF16, device="cuda:0", s=None, m=None, momentum=0.01, seed=None,
                 share=None, head_p=None, head_flags=None, lambda_g=0.0, elastic_std=0.0125, shard=None, elastic_plus=False):
        """shard = (rank, world): class-sharded head -- this replica owns the class columns
        [rank * Cs, (rank + 1) * Cs), Cs = ceil(C / world); the batch of all ranks is gathered for the head, no head
        gradient crosses the wire (frx/ddp.py: sharded_plan).  The backbone stays a plain data-parallel replica."""
        self.kind = HEAD_KINDS[kind] if isinstance(kind, str) else kind
        self.C_full, self.N = num_classes, batch
        self.shard = shard
        if shard is not None:
            rank, world = shard
            self.Cs = -(-num_classes // world)
            self.c0 = rank * self.Cs
            self.C = min(num_classes, self.c0 + self.Cs) - self.c0
            if self.C <= 0:
                raise ValueError(f"class shard {rank} of {world} is empty for {num_classes} classes")
            self.N_g = batch * world
        else:
            self.Cs = self.C = num_classes
            self.c0, self.N_g = 0, batch
        ds, dm = HEAD_DEFAULTS[self.kind]
        self.s, self.m = (ds if s is None else s), (dm if m is None else m)
        self.w_cd = self.kind in ops.W_CD_KINDS
        # (every rank reserves Cs columns, so the flat layouts -- and the backbone's all-reduce ranges -- agree across ranks)
        self.net = ResNet50Engine(batch, dtype, device, extra_params=self.Cs * FEATURE_DIM,
                                  share=None if share is None else share.net)
        self.device = self.net.device
        self.head_p = tuple(HEAD_P_DEFAULTS.get(self.kind, ()) if head_p is None else head_p)
        self.elastic_std = float(elastic_std)
        self.elastic_plus = bool(elastic_plus) and self.kind in ELASTIC_KINDS      # rank-matched margins (criterion.py:1006-1011)
        head_flags = HEAD_FLAG_DEFAULTS.get(self.kind, 0) if head_flags is None else head_flags
        self.head = ops.HeadContext(self.kind, self.N_g, FEATURE_DIM, self.C, self.s, self.m, momentum, device=self.device,
                                    p=self.head_p, flags=head_flags, lambda_g=lambda_g,
                                    class_offset=self.c0 if shard is not None else None)
        if shard is not None:
            dev, ng = self.device, self.N_g
            self.labels_l = torch.zeros(batch, dtype=torch.int64, device=dev)
            self.feats_g = torch.zeros(ng, FEATURE_DIM, device=dev)       # all-gathered features / labels
            self.labels_g = torch.zeros(ng, dtype=torch.int64, device=dev)
            self.ty_g = torch.zeros(ng, device=dev)                       # target cosines (SUM all-reduce)
            self.part = torch.zeros(3, ng, device=dev)                    # row max | sum-exp | rank over the local columns
            self.gmax = torch.zeros(ng, device=dev)                       # row max over ALL columns (MAX all-reduce)
            self.dx_g = torch.zeros(ng, FEATURE_DIM, device=dev)          # this shard's partial dL/dfeats of every row
            self.gout = torch.full((1,), float(world), device=dev)        # the fused SGD rescales every gradient by 1/world
        # Head state (include/frx.h, `state_t`): CurricularFace's `t` [1] (criterion.py:517); AdaFace's batch_mean /
        # batch_std [2] (:838-839), shared between batch sizes like `t`; the elastic heads' per-row margins [N];
        # VPL-ArcFace's class memory `mem` [C,512] followed by `life` [C] (:660-661), shared as well.
        if self.kind == ops.VPL:
            self.t = torch.zeros(num_classes * FEATURE_DIM + num_classes, device=self.device) if share is None else share.t
        elif self.kind in ELASTIC_KINDS:
            self.t = torch.full((batch,), float(self.m), device=self.device)
            self.margin_scratch = torch.empty(batch, device=self.device)
        elif self.kind == ops.ADA:
            self.t = torch.tensor([20.0, 100.0], device=self.device) if share is None else share.t
        else:
            self.t = torch.zeros(1, device=self.device) if share is None else share.t
        self.sphere_iter = 0                                 # SphereFace.iter (criterion.py:33): python int
        self.dfeat = torch.zeros(batch, FEATURE_DIM, device=self.device)
        self.ty_sum = torch.zeros(1, device=self.device)    # sum over the batch of the clamped target cosines (head phase 1)
        self.last = None
        self.world = 1
        self.allreduce = None                                # callable(flat fp32 grads) for data parallel
        self.ty_allreduce = None
        if share is None:
            self.reset_head(seed)

    def set_lambda_g(self, lambda_g):
        """MagFace: weight of loss_g in the total loss the fused backward differentiates (model_utils.py:180)"""
        if self.kind == ops.MAG:
            self.head.desc.lamb = float(lambda_g)

    def sample_margins(self):
        """Elastic heads: this step's per-row margins, normal(m, std) clamped to [m - std, m + std]
        (criterion.py:1002-1004, 1113-1115).  In place on the static buffer (graph replays read it)."""
        if self.kind in ELASTIC_KINDS:
            self.t.normal_(float(self.m), self.elastic_std).clamp_(self.m - self.elastic_std, self.m + self.elastic_std)

    # head weight views, in the reference's own layouts
    def head_w(self, buf=None):
        shape = (self.C, FEATURE_DIM) if self.w_cd else (FEATURE_DIM, self.C)
        return self.net.extra(self.C * FEATURE_DIM, buf).view(shape)

    def reset_head(self, seed=None):
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed + 1)
        C, D = self.C_full, FEATURE_DIM
        if self.kind in (ops.ARC, ops.SPHERE, ops.VPL):      # xavier_uniform_ on [C,D] (criterion.py:244,37,657)
            bound = math.sqrt(6.0 / (C + D))
            w = (torch.rand(C, D, generator=g) * 2 - 1) * bound
        elif self.kind in (ops.MV_AM, ops.MV_ARC):   # same expression on [C,D] (criterion.py:367)
            w = (torch.rand(C, D, generator=g) * 2 - 1).renorm_(2, 1, 1e-5).mul_(1e5)
        elif self.kind in (ops.COS, ops.ADA, ops.MAG):   # uniform(-1,1).renorm_(2,1,1e-5).mul_(1e5) (criterion.py:152,833,1216)
            w = (torch.rand(D, C, generator=g) * 2 - 1).renorm_(2, 1, 1e-5).mul_(1e5)
        else:                                        # normal(std=0.01) (criterion.py:514,973,1080)
            w = torch.randn(D, C, generator=g) * 0.01
        if self.shard is not None:              # the same full-width draw on every rank, each keeps its own columns
            w = w[self.c0:self.c0 + self.C] if self.w_cd else w[:, self.c0:self.c0 + self.C]
        self.head_w().copy_(w.contiguous())
        if self.kind == ops.ADA:
            self.t.copy_(torch.tensor([20.0, 100.0]))
        elif self.kind in ELASTIC_KINDS:
            self.t.fill_(float(self.m))
        else:
            self.t.zero_()
        self.sphere_iter = 0

    def _lamb(self):
        # criterion.py:58-60: iter += 1; lamb = max(5, 1000 * (1 + 0.12*iter)^-1)
        self.sphere_iter += 1
        return max(5.0, 1000.0 * (1 + 0.12 * self.sphere_iter) ** (-1))

    def forward_loss(self, images, labels, want_logits=False, sample=True):
        """sample=False: keep the elastic margins already in the state buffer (a captured graph replays this
        function's kernels only; call sample_margins() before each replay)"""
        if sample:
            self.sample_margins()
        feats = self.net.forward(images)
        lamb = self._lamb() if self.kind == ops.SPHERE else 0.0
        self.last = ops.head_forward(self.head, feats, self.head_w(), labels, state_t=self.t, lamb=lamb,
                                     want_logits=want_logits, ty_allreduce=self.ty_allreduce, elastic_plus=self.elastic_plus)
        self.last["feats"] = feats
        return self.last

    def backward(self, labels, gout=None):
        ops.head_backward(self.head, self.net.feats, self.head_w(), labels, state_t=self.t, gout=gout,
                          dx=self.dfeat, dw=self.head_w(self.net.grads), accumulate_dw=False)
        self.net.backward(self.dfeat)

    # ------------------------------------------------------------------ the step as graph-capturable stages
    # (frx/ddp.py: DataParallelStep drives them; consecutive stages with no exchange between them share one hipGraph)
    #   stage_forward  zero grads, backbone forward, head phase 1 (cosines; sum of target cosines -> self.ty_sum)
    #   [data parallel, CurricularFace only: all-reduce of ty_sum, criterion.py:570-573 on the global batch]
    #   stage_upper    head phase 2 (state update, margin, CE, top-k), head backward, backward of fc + layer4 + layer3
    #   [data parallel: all-reduce of the "upper" gradient ranges starts here and runs under stage_lower]
    #   stage_lower    backward of layer2, layer1, stem
    #   [data parallel: all-reduce of the "lower" ranges; wait for both]
    #   stage_update   fused SGD (grad_scale 1/world, lr from net.lr_dev) + kernel-format weight re-derivation
    exchange_ty = property(lambda self: self.kind == ops.CURR)

    @property
    def flat_grads(self):
        return self.net.grads

    def grad_ranges(self):
        r = self.net.grad_ranges()
        if self.shard is not None:              # the head's columns are this rank's own: their gradient never travels
            cut = self.net.extra_off
            r = {k: [(lo, min(hi, cut)) for lo, hi in v if lo < cut] for k, v in r.items()}
        return r

    def replica_state(self):
        """tensors every data-parallel replica must start equal in (broadcast from rank 0)"""
        net = self.net
        if self.shard is not None:              # (not the head columns: every rank initialised its own slice)
            cut = net.extra_off
            return [net.params[:cut], net.mom[:cut], net.running_mean, net.running_var, net.num_batches_tracked, self.t]
        return [net.params, net.mom, net.running_mean, net.running_var, net.num_batches_tracked, self.t]

    def after_broadcast(self):
        self.net.sync_weights()

    def set_lr(self, lr):
        if getattr(self, "_lr_host", None) != lr:
            self.net.lr_dev.fill_(float(lr))
            self._lr_host = lr

    sgd_momentum, sgd_weight_decay = 0.9, 5e-4      # optim.SGD hyper-parameters of stage_update (model_utils.py:557)

    def graph_key(self):
        """host-side values a captured launch carries BY VALUE: when one changes the graphs are captured again"""
        return (self.head.desc.flags & ~4, self.head.desc.lamb if self.kind == ops.MAG else 0.0,
                self.sgd_momentum, self.sgd_weight_decay, self.world,
                bool(getattr(self.net, "head_bucket", False)), bool(self.net.join_after_upper))

    def pre_step(self):
        """host-side work of a step that stays outside the captured graphs"""
        self.sample_margins()
        if self.kind == ops.SPHERE:                  # lambda of this forward goes through the device state (flags bit 2)
            self.t.fill_(self._lamb())

    def post_replay(self):
        """a graph replay skips the Python bookkeeping of net.forward() / sync_weights(): redo it"""
        owner = self.net.share or self.net
        owner.stats_version = getattr(owner, "stats_version", 0) + 1
        owner.weights_version = getattr(owner, "weights_version", 0) + 1
        self.net._eval_affine_ready = False

    def stage_forward(self, images, labels):
        self.net.training = True
        self.net.ensure_zero_grad()
        feats = self.net.forward(images)
        ops.head_forward_cos(self.head, feats, self.head_w(), labels, state_t=self.t, ty_sum=self.ty_sum)

    def stage_head(self, labels):
        """head phase 2 + head backward + the fc layer's backward: the "head" gradient ranges are final afterwards"""
        if self.elastic_plus:
            ops.rank_matched_margins(self.head, self.t, self.margin_scratch)
        out = ops.head_forward_loss(self.head, labels, self.ty_sum, self.N * self.world, state_t=self.t,
                                    lamb=None if self.kind == ops.SPHERE else 0.0)
        out["feats"] = self.net.feats
        self.last = out
        ops.head_backward(self.head, self.net.feats, self.head_w(), labels, state_t=self.t, dx=self.dfeat,
                          dw=self.head_w(self.net.grads), accumulate_dw=False)
        self.net.backward_fc(self.dfeat)
        return out

    def stage_upper_rest(self):
        self.net.backward_upper_rest()

    def stage_upper(self, labels):
        out = self.stage_head(labels)
        self.stage_upper_rest()
        return out

    def set_head_bucket(self, on):
        self.net.set_head_bucket(on)

    # ---- class-sharded head: the compute between the collectives of frx/ddp.py: sharded_plan
    def shard_stage_backbone(self, images, labels):
        self.net.training = True
        self.net.ensure_zero_grad()
        self.net.forward(images)
        self.feats_l = self.net.feats
        self.labels_l.copy_(labels)

    def shard_stage_cos(self):
        ops.head_shard_cos(self.head, self.feats_g, self.head_w(), self.labels_g, self.ty_g)

    def shard_stage_rows(self):
        if self.kind == ops.SPHERE:
            self.head.desc.flags |= 4
        ops.head_shard_rows(self.head, self.labels_g, self.ty_g, self.part, state_t=self.t)
        self.gmax.copy_(self.part[0])

    def shard_stage_rescale(self):
        ops.head_shard_rescale(self.part[0], self.gmax, self.part[1])

    def shard_stage_head_bwd(self):
        out = ops.head_shard_finish(self.head, self.gmax, self.part[1], self.part[2], state_t=self.t)
        out["feats"] = self.net.feats
        self.last = out
        ops.head_backward(self.head, self.feats_g, self.head_w(), self.labels_g, state_t=self.t, gout=self.gout,
                          dx=self.dx_g, dw=self.head_w(self.net.grads), accumulate_dw=False)
        return out

    def shard_stage_upper(self):
        self.net.backward_upper(self.dfeat)

    def gather_head_weight(self, group=None):
        """the full [C, 512] / [512, C] head weight assembled from every rank's columns (checkpoints)"""
        import torch.distributed as dist
        world = self.shard[1]
        mine = torch.zeros(self.Cs, FEATURE_DIM, device=self.device)
        mine[:self.C] = self.head_w() if self.w_cd else self.head_w().t()
        full = torch.empty(world * self.Cs, FEATURE_DIM, device=self.device)
        dist.all_gather_into_tensor(full, mine, group=group)
        full = full[:self.C_full]
        return full.contiguous() if self.w_cd else full.t().contiguous()

    def stage_lower(self):
        self.net.backward_lower()

    def stage_update(self):
        self.net.sgd_step(None, self.sgd_momentum, self.sgd_weight_decay, grad_scale=1.0 / self.world)

    def train_step(self, images, labels, lr=None):
        """zero_grad -> forward -> CE -> backward -> [all-reduce] -> SGD (model_utils.py:176-187), eagerly."""
        self.net.training = True
        self.net.ensure_zero_grad()
        out = self.forward_loss(images, labels)
        self.backward(labels)
        if self.allreduce is not None:
            self.allreduce(self.net.grads)
        self.net.sgd_step(lr, grad_scale=1.0 / self.world)
        return out

    @torch.no_grad()
    def embed(self, images):
        self.net.training = False
        return self.net.forward(images)
