"""Margin heads and model wrappers with the reference's class names, constructor signatures,
attribute names (= state-dict keys) and forward contract (main_code/utils/criterion.py:12-135,
137-230, 232-325, 491-617).  The arithmetic runs in libfrx (frx/ops.py, csrc/head.hip); these
classes hold parameters and per-head state only.

Training-mode forward of a *Net returns ([cos_s, logits], norms, loss_g, one_hot) -- `logits`
carries autograd back into the native backward; eval-mode forward returns raw features [N, 512]."""
import math

import torch
import torch.nn as nn

from frx import ops
from frx.module import NativeFaceNet

from .backbones import get_backbone
from .config import (COMPUTE_DTYPE, FEATURE_DIM, M_arc, M_cos, M_curricular, M_sphere, MOMENTUM_curricular, S_arc,
                     S_cos, S_curricular)


class _HeadBase(nn.Module):
    """Parameter/state holder; calling it stand-alone on features runs the native head too."""
    kind = ops.ARC

    def _param(self):
        return self.weight if hasattr(self, "weight") else self.kernel

    def forward(self, feats, labels):
        return _StandaloneHead.run(self, feats, labels)


class SphereFace(_HeadBase):
    """criterion.py:12-107.  m=2 only (config.py:17); `iter` is a Python int as upstream (:33), so it is not
    part of the state dict and restarts on resume."""
    kind = ops.SPHERE

    def __init__(self, in_features, out_features, device_id=None, m=4):
        super().__init__()
        if device_id is not None:
            raise NotImplementedError("the reference's dormant model-parallel branch (device_id) is not supported")
        if int(m) != 2:
            raise NotImplementedError("native SphereFace implements m=2 (utils.config.M_sphere)")
        self.in_features, self.num_classes, self.out_features = in_features, out_features, out_features
        self.m, self.s, self.device_id = int(m), 1.0, None
        self.base, self.gamma, self.power, self.LambdaMin, self.iter, self.lamb = 1000.0, 0.12, 1, 5.0, 0, 0.0
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)

    def get_proxy(self, labels):
        return self.weight.permute(1, 0)[:, labels].clone().detach()


class CosFace(_HeadBase):
    """criterion.py:137-197; parameter `kernel` is [D, C]."""
    kind = ops.COS

    def __init__(self, embedding_size=512, classnum=51332, s=64.0, m=0.4):
        super().__init__()
        self.classnum = self.num_classes = classnum
        self.s, self.m, self.eps = s, m, 1e-4
        self.kernel = nn.Parameter(torch.empty(embedding_size, classnum))
        self.kernel.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)

    def get_proxy(self, labels):
        return self.kernel[:, labels].clone().detach()


class ArcFace(_HeadBase):
    """criterion.py:232-301; parameter `weight` is [C, D]; only easy_margin=False (what ArcFaceNet passes,
    :313) has a native epilogue."""
    kind = ops.ARC

    def __init__(self, embed_size, num_classes, device_id=None, s=64.0, m=0.50, easy_margin=True):
        super().__init__()
        if device_id is not None:
            raise NotImplementedError("the reference's dormant model-parallel branch (device_id) is not supported")
        if easy_margin:
            raise NotImplementedError("native ArcFace implements easy_margin=False (as ArcFaceNet constructs it)")
        self.in_features, self.out_features, self.num_classes = embed_size, num_classes, num_classes
        self.s, self.m, self.easy_margin, self.device_id = s, m, False, None
        self.weight = nn.Parameter(torch.empty(num_classes, embed_size))
        nn.init.xavier_uniform_(self.weight)
        self.cos_m, self.sin_m = math.cos(m), math.sin(m)
        self.th, self.mm = math.cos(math.pi - m), math.sin(math.pi - m) * m

    def get_proxy(self, labels):
        return self.weight.t()[:, labels].clone().detach()      # [D, N] (upstream :258 indexes the wrong axis)


class CurricularFace(_HeadBase):
    """criterion.py:491-587; parameter `kernel` [D, C], buffer `t` [1] (EMA of the target cosine)."""
    kind = ops.CURR

    def __init__(self, feat_dim, num_class, m=0.5, s=64.0, momentum=0.01):
        super().__init__()
        self.num_classes, self.m, self.s, self.momentum = num_class, m, s, momentum
        self.cos_m, self.sin_m = math.cos(m), math.sin(m)
        self.threshold, self.mm = math.cos(math.pi - m), math.sin(math.pi - m) * m
        self.kernel = nn.Parameter(torch.empty(feat_dim, num_class))
        nn.init.normal_(self.kernel, std=0.01)
        self.register_buffer("t", torch.zeros(1))

    def get_proxy(self, labels):
        return self.kernel[:, labels].clone().detach()


class _StandaloneHead(torch.autograd.Function):
    """head(feats, labels) outside a *Net (feature tensors from anywhere)."""

    @staticmethod
    def run(head, feats, labels):
        if not feats.is_cuda:
            raise ops.FrxError("native heads run on a HIP device only (no CPU fallback)")
        w = head._param()
        N, D = feats.shape
        ctx = ops.HeadContext(head.kind, N, D, head.num_classes, head.s, float(head.m), getattr(head, "momentum", 0.01),
                              device=feats.device)
        lamb = 0.0
        if head.kind == ops.SPHERE:
            head.iter += 1
            head.lamb = lamb = max(head.LambdaMin, head.base * (1 + head.gamma * head.iter) ** (-head.power))
        x = feats.detach().float().contiguous()
        t = head.t if hasattr(head, "t") else None
        out = ops.head_forward(ctx, x, w.detach().contiguous(), labels.contiguous(), state_t=t, lamb=lamb, want_logits=True)
        logits = _StandaloneHead.apply(feats, w, out["logits"], ctx, x, labels.contiguous(), t)
        one_hot = torch.zeros_like(out["cos_s"]).scatter_(1, labels.view(-1, 1), 1.0)
        return [out["cos_s"], logits], out["norms"].view(-1, 1), 0, one_hot

    @staticmethod
    def forward(ctx, feats, w, logits, hctx, x, labels, t):
        ctx.hctx, ctx.x, ctx.labels, ctx.t = hctx, x, labels, t
        ctx.save_for_backward(w)
        return logits.view_as(logits)

    @staticmethod
    def backward(ctx, dlogits):
        (w,) = ctx.saved_tensors
        dx, dw = ops.head_backward_dlogits(ctx.hctx, ctx.x, w.detach().contiguous(), ctx.labels,
                                           dlogits.contiguous().float(), state_t=ctx.t)
        return dx, dw, None, None, None, None, None


def _net(name, kind, attr, make_head):
    class Net(NativeFaceNet):
        head_attr = attr

        def __init__(self, num_classes, backbone):
            super().__init__(make_head(num_classes), get_backbone(backbone), COMPUTE_DTYPE)
            self.loss_model = name
    Net.kind = kind
    return Net


class SphereFaceNet(_net("sphereface", ops.SPHERE, "sphereface", lambda c: SphereFace(FEATURE_DIM, c, m=M_sphere))):
    """criterion.py:109-135"""


class CosFaceNet(_net("cosface", ops.COS, "cosface", lambda c: CosFace(FEATURE_DIM, c, s=S_cos, m=M_cos))):
    """criterion.py:199-230"""


class ArcFaceNet(_net("arcface", ops.ARC, "arcface", lambda c: ArcFace(FEATURE_DIM, c, s=S_arc, m=M_arc, easy_margin=False))):
    """criterion.py:303-325"""


class CurricularFaceNet(_net("curricularface", ops.CURR, "curricular",
                             lambda c: CurricularFace(FEATURE_DIM, c, m=M_curricular, s=S_curricular, momentum=MOMENTUM_curricular))):
    """criterion.py:589-617"""
