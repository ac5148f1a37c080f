"""nn.Module facade over the native engine, so the reference's object model keeps working:

    model = ArcFaceNet(num_classes, backbone).to(device); model.train() / .eval()
    (cos_s, logits), norms, loss_g, one_hot = model(images, labels)     # training
    feats = model(images)                                               # eval
    model.state_dict() / load_state_dict() with the reference's key names
    torch.optim.SGD(model.parameters(), ...)                            # a foreign optimiser still works

Parameters are registered as ordinary nn.Parameters (torchvision / reference names and shapes).
They start on the CPU; the first forward on a HIP device builds the engine, copies them in and then
RE-POINTS each Parameter at its slice of the engine's flat fp32 buffer (conv weights as
channels-last-strided views of the KRSC master copy), so optimiser updates land directly in engine
memory.  A version check re-derives the bf16 kernel-format weights when something else modified them.

There is no CPU execution path: forward on a CPU tensor raises FrxError.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ddp
from . import engine as E
from . import ops
from ._lib import FrxError

_DTYPES = {"bf16": ops.BF16, "f32": ops.F32, "fp32": ops.F32, "float32": ops.F32, "bfloat16": ops.BF16}


class NativeBackbone(nn.Module):
    """ResNet-50 parameter container with torchvision's state-dict layout (backbones.py:16-18 upstream:
    torchvision resnet50 + nn.Linear(2048, FEATURE_DIM)).  Holds no arithmetic of its own."""

    def __init__(self, feature_dim=E.FEATURE_DIM):
        super().__init__()
        if feature_dim != E.FEATURE_DIM:
            raise ValueError(f"native backbone emits {E.FEATURE_DIM}-d features")
        stem, blocks, _ = E.resnet50_specs()
        self._specs = [stem] + [c for b in blocks for c in (b.conv1, b.conv2, b.conv3, b.down) if c is not None]
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        for c in self._specs:
            k = 7 if c.stem else c.k
            w = torch.randn(c.Co, c.Ci, k, k, generator=g) * math.sqrt(2.0 / (c.Co * k * k))   # kaiming fan_out
            self._reg(c.name + ".weight", nn.Parameter(w))
            self._reg(c.bn + ".weight", nn.Parameter(torch.ones(c.Co)))
            self._reg(c.bn + ".bias", nn.Parameter(torch.zeros(c.Co)))
            self._reg(c.bn + ".running_mean", torch.zeros(c.Co), buffer=True)
            self._reg(c.bn + ".running_var", torch.ones(c.Co), buffer=True)
            self._reg(c.bn + ".num_batches_tracked", torch.zeros((), dtype=torch.long), buffer=True)
        bound = 1.0 / math.sqrt(2048)
        self._reg("fc.weight", nn.Parameter((torch.rand(feature_dim, 2048, generator=g) * 2 - 1) * bound))
        self._reg("fc.bias", nn.Parameter((torch.rand(feature_dim, generator=g) * 2 - 1) * bound))

    def _reg(self, dotted, tensor, buffer=False):
        """register under nested sub-modules so state_dict keys read 'layer1.0.conv1.weight'"""
        mod = self
        *path, leaf = dotted.split(".")
        for p in path:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        if buffer:
            mod.register_buffer(leaf, tensor)
        else:
            mod.register_parameter(leaf, tensor)

    def forward(self, x):
        raise FrxError("NativeBackbone is driven through its owning *Net module (no standalone forward)")


class NativeFaceNet(nn.Module):
    """Base of ArcFaceNet / CosFaceNet / SphereFaceNet / CurricularFaceNet."""

    head_attr = "head"          # attribute name of the head module == state-dict prefix
    kind = ops.ARC

    def __init__(self, head_module, backbone_module, compute_dtype="bf16"):
        super().__init__()
        self.backbone = backbone_module
        setattr(self, self.head_attr, head_module)
        self._dtype = _DTYPES[compute_dtype]
        self._engines = {}            # batch size -> FaceEngine (all sharing the first one's parameters)
        self._primary = None
        self._synced_version = None
        self._last_ctx = None
        self._steppers = {}           # (batch size, input dtype) -> ddp.DataParallelStep (the fused train step)
        self._dp = None               # data_parallel() settings
        self.use_graph = True         # fused train step: replay hipGraphs from the second step of a batch size on

    # ------------------------------------------------------------------ engine management
    @property
    def head(self):
        return self._modules[self.head_attr]      # (the elastic *Nets name the attribute itself `head`)

    def _head_param(self):
        return self.head.weight if hasattr(self.head, "weight") else self.head.kernel

    def _engine_for(self, n, device):
        if device.type != "cuda":
            raise FrxError("the native face-recognition path runs on a HIP device only: move the model and the "
                           "batch with .to('cuda') (there is no CPU fallback)")
        if self._primary is not None and self._primary.device != device:
            raise FrxError(f"model engine lives on {self._primary.device}, batch is on {device}")
        eng = self._engines.get(n)
        if eng is not None:
            return eng
        h = self.head
        eng = E.FaceEngine(self.kind, h.num_classes, n, dtype=self._dtype, device=device, s=h.s, m=float(h.m),
                           momentum=getattr(h, "momentum", 0.01), share=self._primary,
                           head_p=getattr(h, "frx_p", None), head_flags=getattr(h, "frx_flags", 0),
                           elastic_std=getattr(h, "std", 0.0125), elastic_plus=bool(getattr(h, "plus", False)))
        if self._primary is None:
            self._adopt(eng)
            if self._dp is not None:
                ddp.broadcast_parameters(eng, 0, self._dp["group"])
        if self._dp is not None:
            import torch.distributed as dist
            self._attach_dp(eng, dist.get_world_size(self._dp["group"]))
        self._engines[n] = eng
        return eng

    # ------------------------------------------------------------------ fused step / data parallel
    def data_parallel(self, group=None, bf16_buckets=False):
        """Make this model one replica of a data-parallel job (torch.distributed must be initialised, one process per
        GPU).  Rank 0's parameters are broadcast when the engine is built; the fused train step (train_model with
        FusedSGD) all-reduces gradients overlapped with the backward (frx/ddp.py); the autograd-compatible path
        all-reduces after its backward.  The reference has no multi-GPU path."""
        import torch.distributed as dist
        self._dp = dict(group=group, bf16_buckets=bool(bf16_buckets))
        world = dist.get_world_size(group)
        for eng in self._engines.values():
            self._attach_dp(eng, world)
        if self._primary is not None:
            ddp.broadcast_parameters(self._primary, 0, group)
        return self

    def _attach_dp(self, eng, world):
        import torch.distributed as dist
        group = self._dp["group"]
        eng.world = world

        def allreduce(flat):
            dist.all_reduce(flat, group=group)

        def ty_allreduce(t):
            dist.all_reduce(t, group=group)
            return eng.N * world
        eng.allreduce = allreduce if world > 1 else None
        eng.ty_allreduce = ty_allreduce if (world > 1 and eng.exchange_ty) else None

    def _stepper_for(self, eng, images):
        key = (eng.N, images.dtype)
        st = self._steppers.get(key)
        if st is None:
            dp = self._dp or {}
            st = ddp.DataParallelStep(eng, group=dp.get("group"), use_graph=self.use_graph,
                                      bf16_buckets=dp.get("bf16_buckets", False), broadcast=False)
            self._steppers[key] = st
        return st

    def _sync_head_flags(self, eng):
        """heads whose flags change at run time (VPL-ArcFace's change_training_mode, criterion.py:676-679)"""
        f = getattr(self.head, "frx_flags", None)
        if f is not None:
            eng.head.desc.flags = int(f)

    def _adopt(self, eng):
        """first engine: load the module's current tensors, then alias every Parameter / buffer to it"""
        net = eng.net
        sd = {k: v.detach() for k, v in self.backbone.state_dict().items()}
        net.load_state_dict(sd)
        eng.head_w().copy_(self._head_param().detach().to(eng.device))
        if eng.kind == ops.CURR:
            eng.t.copy_(self.head.t.detach().to(eng.device))
        elif eng.kind == ops.ADA:
            eng.t.copy_(torch.cat([self.head.batch_mean.detach().view(1), self.head.batch_std.detach().view(1)]).to(eng.device))
        elif eng.kind == ops.VPL:
            eng.t.copy_(torch.cat([self.head.mem.detach().reshape(-1), self.head.life.detach().reshape(-1)]).to(eng.device))
        mods = dict(self.backbone.named_modules())

        def sub(dotted):
            path, leaf = dotted.rsplit(".", 1)
            return mods[path], leaf
        for i, c in enumerate(net.convs):
            w = net.w_master(c)
            g = net.w_grad(c)
            mo = net.w_grad(c, net.mom)
            if c.stem:
                w, g, mo = w[:, :, :7, :3], g[:, :, :7, :3], mo[:, :, :7, :3]
            m, leaf = sub(c.name + ".weight")
            self._alias(m, leaf, w.permute(0, 3, 1, 2), g.permute(0, 3, 1, 2), mo.permute(0, 3, 1, 2))
            m, _ = sub(c.bn + ".weight")
            self._alias(m, "weight", net.gamma(c), net.gamma(c, net.grads), net.gamma(c, net.mom))
            self._alias(m, "bias", net.beta(c), net.beta(c, net.grads), net.beta(c, net.mom))
            m.running_mean = net._bn(net.running_mean, c)
            m.running_var = net._bn(net.running_var, c)
            m.num_batches_tracked = net.num_batches_tracked[i]
        self._alias(self.backbone.fc, "weight", net.fc_w(), net.fc_w(net.grads), net.fc_w(net.mom))
        self._alias(self.backbone.fc, "bias", net.fc_b(), net.fc_b(net.grads), net.fc_b(net.mom))
        pname = "weight" if hasattr(self.head, "weight") else "kernel"
        self._alias(self.head, pname, eng.head_w(), eng.head_w(net.grads), eng.head_w(net.mom))
        if eng.kind == ops.CURR:
            self.head.t = eng.t
        elif eng.kind == ops.ADA:                      # the two EMA buffers live in the engine's head state
            self.head.batch_mean, self.head.batch_std = eng.t[0:1], eng.t[1:2]
            self.head.t = self.head.t.to(eng.device)
        elif eng.kind == ops.VPL:                      # class memory and its life counters live in the engine's head state
            cd = eng.C * E.FEATURE_DIM
            self.head.mem, self.head.life = eng.t[:cd].view(eng.C, E.FEATURE_DIM), eng.t[cd:]
            for k in ("cos_m", "sin_m", "th", "mm"):
                setattr(self.head, k, getattr(self.head, k).to(eng.device))
        self._primary = eng
        self._param_list = list(self.parameters())
        self._synced_version = self._version_sum()

    def _alias(self, module, leaf, view, grad_view, mom_view):
        p = getattr(module, leaf)
        p.data = view                     # same storage as the engine's flat buffer from now on
        p._frx_grad = grad_view
        p._frx_mom = mom_view             # the SGD momentum of this parameter, in the parameter's own (torch) shape

    def _version_sum(self):
        # a Parameter re-pointed with `.data = view` keeps its OWN version counter, bumped by every
        # in-place write through it (torch optimiser step, load_state_dict's copy_)
        return sum(p._version for p in self._param_list)

    def _resync_if_touched(self):
        """Parameters were modified in place through the nn.Parameter aliases: re-derive the
        kernel-format (bf16 KRSC / CRSK) copies before they are used."""
        v = self._version_sum()
        if v != self._synced_version:
            self._primary.net.sync_weights()
            self._synced_version = v

    # ------------------------------------------------------------------ nn.Module surface
    def _apply(self, fn, *a, **k):
        if self._primary is not None:
            probe = torch.empty(0, device=self._primary.device)
            moved = fn(probe)
            if moved.device == probe.device and moved.dtype == probe.dtype:
                return self                   # .to(same device) / .float(): nothing to do
            raise FrxError("the model is bound to its HIP engine; move it with .to(device) BEFORE the first forward")
        return super()._apply(fn, *a, **k)

    def state_dict(self, *args, **kwargs):
        """reference key names; tensors are contiguous copies (the live parameters are strided views
        into the engine's flat buffer)"""
        sd = super().state_dict(*args, **kwargs)
        if kwargs.get("keep_vars", False):
            return sd
        for k in list(sd.keys()):
            sd[k] = sd[k].detach().clone().contiguous()
        return sd

    def forward(self, x, labels=None):
        eng = self._engine_for(x.shape[0], x.device)
        self._resync_if_touched()
        self._sync_head_flags(eng)
        if not self.training:
            with torch.no_grad():
                eng.net.training = False
                return eng.net.forward(x.contiguous()).clone()          # raw feats [N,512], like the reference
        assert labels is not None
        return _TrainForward.run(self, eng, x, labels)

    # grads: autograd's zero_grad(set_to_none=True) drops .grad; the engine's flat gradient buffer is the
    # real accumulator, so "all grads are None" means "logically zero" and is applied before accumulating
    def _begin_backward(self):
        eng = self._primary
        params = list(self.parameters())
        if all(p.grad is None for p in params):
            eng.net.grads.zero_()
        return params

    def _publish_grads(self, params):
        for p in params:
            if p.grad is None:
                p.grad = p._frx_grad


class _TrainForward(torch.autograd.Function):
    """logits = F(images): forward runs backbone + head on the engine; backward receives dL/dlogits from
    whatever criterion the caller applied (model_utils.py:178-185) and replays the native backward."""

    @staticmethod
    def run(model, eng, x, labels):
        labels = labels.contiguous()
        h = model.head
        eng.net.training = True
        with torch.no_grad():
            lamb = 0.0
            if eng.kind == ops.SPHERE:               # SphereFace.iter lives on the module (criterion.py:33,58-60)
                h.iter += 1
                h.lamb = max(h.LambdaMin, h.base * (1 + h.gamma * h.iter) ** (-h.power))
                lamb = h.lamb
            eng.sample_margins()                       # elastic heads: this step's margins (criterion.py:1002,1113)
            feats = eng.net.forward(x.contiguous())
            out = ops.head_forward(eng.head, feats, eng.head_w(), labels, state_t=eng.t, lamb=lamb, want_logits=True,
                                   ty_allreduce=eng.ty_allreduce, elastic_plus=eng.elastic_plus)
        anchor = model._head_param()
        is_mag = eng.kind == ops.MAG
        logits, loss_g = _TrainForward.apply(anchor, out["logits"], out["loss_g"][0] if is_mag else None, model, eng, labels)
        one_hot = torch.zeros_like(out["cos_s"]).scatter_(1, labels.view(-1, 1), 1.0)
        model._last_ctx = out
        # MagFace returns the clamped x_norm and a differentiable loss_g (criterion.py:1291); the others norms and 0
        return [out["cos_s"], logits], out["norms"].view(-1, 1), (loss_g if is_mag else 0), one_hot

    @staticmethod
    def forward(ctx, anchor, logits, loss_g, model, eng, labels):
        ctx.model, ctx.eng, ctx.labels = model, eng, labels
        ctx.set_materialize_grads(False)
        return logits.view_as(logits), (None if loss_g is None else loss_g.view_as(loss_g))

    @staticmethod
    def backward(ctx, dlogits, dloss_g):
        model, eng, labels = ctx.model, ctx.eng, ctx.labels
        params = model._begin_backward()
        if dlogits is None:
            dlogits = torch.zeros(eng.N, eng.C, device=eng.device)
        if eng.kind == ops.MAG:                         # upstream dL/dloss_g (= args.lambda_g, model_utils.py:180)
            eng.head.desc.lamb = 0.0 if dloss_g is None else float(dloss_g)
        ops.head_backward_dlogits(eng.head, eng.net.feats, eng.head_w(), labels, dlogits.contiguous().float(),
                                  state_t=eng.t, dx=eng.dfeat, dw=eng.head_w(eng.net.grads), accumulate_dw=True)
        eng.net.backward(eng.dfeat)
        if eng.allreduce is not None:
            eng.allreduce(eng.net.grads)
            eng.net.grads.mul_(1.0 / eng.world)
        model._publish_grads(params)
        return None, None, None, None, None, None
