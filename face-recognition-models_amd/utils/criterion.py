"""Margin heads and model wrappers with the reference's class names, constructor signatures,
attribute names (= state-dict keys) and forward contract (main_code/utils/criterion.py:12-135,
137-230, 232-325, 491-617; and, SURVEY 8(f)-3, MV_Softmax :327-489, AdaFace :795-949, ElasticCosFace :951-1052,
ElasticArcFace :1054-1176, MagFace :1178-1329, VPLArcFace :619-793).  The arithmetic runs in libfrx (frx/ops.py, csrc/head.hip); these
classes hold parameters and per-head state only.

Training-mode forward of a *Net returns ([cos_s, logits], norms, loss_g, one_hot) -- `logits`
carries autograd back into the native backward; eval-mode forward returns raw features [N, 512]."""
import math

import torch
import torch.nn as nn

from frx import ops
from frx.module import NativeFaceNet

from .backbones import get_backbone
from .config import (COMPUTE_DTYPE, DELTA_vpl, EASY_MARGIN_mag, EASY_MARGIN_vpl, LAMDA_vpl, M_vpl, S_vpl, FEATURE_DIM, H_ada, L_A_mag, L_MARGIN_mag, M_ada, M_arc, M_cos,
                     M_curricular, M_elastic_arc, M_elastic_cos, M_mv, M_sphere, MARGIN_TYPE_mv, MOMENTUM_curricular,
                     PLUS_elastic_arc, PLUS_elastic_cos, S_ada, S_arc, S_cos, S_curricular, S_elastic_arc, S_elastic_cos,
                     S_mag, S_mv, STD_elastic_arc, STD_elastic_cos, T_ALPHA_ada, U_A_mag, U_MARGIN_mag, WEIGHT_mv)


class _HeadBase(nn.Module):
    """Parameter/state holder; calling it stand-alone on features runs the native head too."""
    kind = ops.ARC

    def _param(self):
        return self.weight if hasattr(self, "weight") else self.kernel

    def forward(self, feats, labels):
        return _StandaloneHead.run(self, feats, labels)


class SphereFace(_HeadBase):
    """criterion.py:12-107.  m = 1..5 (the Chebyshev table :40-47; config.py:17 ships 2); `iter` is a Python int as
    upstream (:33), so it is not part of the state dict and restarts on resume."""
    kind = ops.SPHERE

    def __init__(self, in_features, out_features, device_id=None, m=4):
        super().__init__()
        if device_id is not None:
            raise NotImplementedError("the reference's dormant model-parallel branch (device_id) is not supported")
        if int(m) not in (1, 2, 3, 4, 5):
            raise ValueError(f"SphereFace margin m={m}: the Chebyshev table (criterion.py:40-47) holds m = 0..5, and m = 0 is no margin")
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
    """criterion.py:232-301; parameter `weight` is [C, D]; easy_margin (:284-285) is a flag of the native epilogue
    (ArcFaceNet passes False, :313)."""
    kind = ops.ARC

    def __init__(self, embed_size, num_classes, device_id=None, s=64.0, m=0.50, easy_margin=True):
        super().__init__()
        if device_id is not None:
            raise NotImplementedError("the reference's dormant model-parallel branch (device_id) is not supported")
        self.in_features, self.out_features, self.num_classes = embed_size, num_classes, num_classes
        self.s, self.m, self.easy_margin, self.device_id = s, m, bool(easy_margin), None
        self.frx_flags = 1 if easy_margin else 0
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


def _no_model_parallel(device_id):
    if device_id is not None:
        raise NotImplementedError("the reference's dormant model-parallel branch (device_id) is not supported")


class MV_Softmax(_HeadBase):
    """criterion.py:327-450; parameter `weight` [C, D]; margin_type 'am' (CosFace-style) or 'arc'."""

    def __init__(self, feat_dim, num_class, margin=0.35, mv_weight=1.12, s=32.0, margin_type='arc', device_id=None):
        super().__init__()
        _no_model_parallel(device_id)
        self.margin_type = margin_type.lower()
        assert self.margin_type in ('am', 'arc'), "margin_type must be 'am' or 'arc'"
        self.kind = ops.MV_AM if self.margin_type == 'am' else ops.MV_ARC
        self.feat_dim, self.num_class, self.num_classes = feat_dim, num_class, num_class
        self.margin = self.m = margin
        self.mv_weight, self.s, self.device_id = mv_weight, s, None
        self.frx_p = (mv_weight,)
        self.weight = nn.Parameter(torch.empty(num_class, feat_dim))
        self.weight.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)
        if self.margin_type == 'arc':
            self.cos_m, self.sin_m = math.cos(margin), math.sin(margin)
            self.th, self.mm = math.cos(math.pi - margin), math.sin(margin) * margin

    def get_proxy(self, labels):
        return self.weight.t()[:, labels].clone().detach()      # [D, N] (upstream :386 indexes the wrong axis)


class AdaFace(_HeadBase):
    """criterion.py:795-907; parameter `kernel` [D, C]; buffers `t` (unused upstream), `batch_mean`, `batch_std`."""
    kind = ops.ADA

    def __init__(self, feat_dim, num_class, m=0.4, h=0.333, s=64.0, t_alpha=1.0, device_id=None):
        super().__init__()
        _no_model_parallel(device_id)
        self.feat_dim, self.num_class, self.num_classes = feat_dim, num_class, num_class
        self.m, self.h, self.s, self.t_alpha, self.eps, self.device_id = m, h, s, t_alpha, 1e-3, None
        self.frx_p = (h, t_alpha)
        self.kernel = nn.Parameter(torch.empty(feat_dim, num_class))
        self.kernel.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)
        self.register_buffer('t', torch.zeros(1))
        self.register_buffer('batch_mean', torch.ones(1) * 20)
        self.register_buffer('batch_std', torch.ones(1) * 100)

    def get_proxy(self, labels):
        return self.kernel[:, labels].clone().detach()


class _Elastic(_HeadBase):
    """criterion.py:951-1021 / 1054-1145; parameter `kernel` [D, C].  The per-row margin is drawn on the device each
    step (normal(m, std) clamped to [m - std, m + std]); with `plus` the sorted margins are then assigned by the rank of
    the rows' target cosines (:1006-1011, :1117-1122) between the two head phases (ops.rank_matched_margins)."""

    def __init__(self, feat_dim, num_class, s=64.0, m=0.35, std=0.0125, plus=False, device_id=None):
        super().__init__()
        _no_model_parallel(device_id)
        self.feat_dim, self.num_class, self.num_classes = feat_dim, num_class, num_class
        self.s, self.m, self.std, self.plus, self.device_id = s, m, std, bool(plus), None
        self.kernel = nn.Parameter(torch.empty(feat_dim, num_class))
        nn.init.normal_(self.kernel, std=0.01)

    def get_proxy(self, labels):
        return self.kernel[:, labels].clone().detach()


class ElasticCosFace(_Elastic):
    kind = ops.ELASTIC_COS


class ElasticArcFace(_Elastic):
    kind = ops.ELASTIC_ARC

    def __init__(self, feat_dim, num_class, s=64.0, m=0.50, std=0.0125, plus=False, device_id=None):
        super().__init__(feat_dim, num_class, s, m, std, plus, device_id)


class MagFace(_HeadBase):
    """criterion.py:1178-1291; parameter `kernel` [D, C].  forward returns the CLAMPED norms and loss_g (:1291)."""
    kind = ops.MAG

    def __init__(self, feat_dim, num_class, s=64.0, easy_margin=True, l_margin=0.45, u_margin=0.8, l_a=10.0, u_a=110.0,
                 device_id=None):
        super().__init__()
        _no_model_parallel(device_id)
        self.feat_dim, self.num_class, self.num_classes = feat_dim, num_class, num_class
        self.s, self.m, self.easy_margin, self.device_id = s, 0.0, easy_margin, None
        self.l_margin, self.u_margin, self.l_a, self.u_a = l_margin, u_margin, l_a, u_a
        self.frx_p, self.frx_flags = (l_margin, u_margin, l_a, u_a), int(bool(easy_margin))
        self.kernel = nn.Parameter(torch.empty(feat_dim, num_class))
        self.kernel.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)

    def get_proxy(self, labels):
        return self.kernel[:, labels].clone().detach()

    def _margin(self, x_norm):
        return (self.u_margin - self.l_margin) / (self.u_a - self.l_a) * (x_norm - self.l_a) + self.l_margin

    def calc_loss_G(self, x_norm):
        return torch.mean(1 / (self.u_a ** 2) * x_norm + 1 / x_norm)


class VPLArcFace(_HeadBase):
    """criterion.py:619-752; parameter `weight` [C, D]; buffers `mem` [C, D], `life` [C] (the class memory) and the
    four ArcFace constants upstream registers as buffers (:664-667)."""
    kind = ops.VPL

    def __init__(self, feat_dim, num_class, s=64.0, m=0.50, easy_margin=True, lamda=0.15, delta=100, device_id=None):
        super().__init__()
        _no_model_parallel(device_id)
        self.feat_dim, self.num_class, self.num_classes = feat_dim, num_class, num_class
        self.s, self.m, self.easy_margin, self.lamda, self.delta, self.device_id = s, m, easy_margin, lamda, delta, None
        self.frx_p = (lamda, float(delta))
        self.weight = nn.Parameter(torch.empty(num_class, feat_dim))
        nn.init.xavier_uniform_(self.weight)
        self.register_buffer('mem', torch.zeros(num_class, feat_dim))
        self.register_buffer('life', torch.zeros(num_class))
        self.register_buffer('cos_m', torch.tensor(math.cos(m), dtype=torch.float32))
        self.register_buffer('sin_m', torch.tensor(math.sin(m), dtype=torch.float32))
        self.register_buffer('th', torch.tensor(math.cos(math.pi - m), dtype=torch.float32))
        self.register_buffer('mm', torch.tensor(math.sin(math.pi - m) * m, dtype=torch.float32))
        self.norm_training_flag = True

    @property
    def frx_flags(self):
        return int(bool(self.easy_margin)) | (2 if self.norm_training_flag else 0)

    def change_training_mode(self, flag):
        self.norm_training_flag = flag

    def get_proxy(self, labels):
        return self.weight.t()[:, labels].clone().detach()      # [D, N] (upstream :684 indexes the wrong axis)


class _StandaloneHead(torch.autograd.Function):
    """head(feats, labels) outside a *Net (feature tensors from anywhere)."""

    @staticmethod
    def run(head, feats, labels):
        if not feats.is_cuda:
            raise ops.FrxError("native heads run on a HIP device only (no CPU fallback)")
        w = head._param()
        N, D = feats.shape
        kind = head.kind
        ctx = ops.HeadContext(kind, N, D, head.num_classes, head.s, float(head.m), getattr(head, "momentum", 0.01),
                              device=feats.device, p=getattr(head, "frx_p", ()), flags=getattr(head, "frx_flags", 0))
        lamb = 0.0
        if kind == ops.SPHERE:
            head.iter += 1
            head.lamb = lamb = max(head.LambdaMin, head.base * (1 + head.gamma * head.iter) ** (-head.power))
        x = feats.detach().float().contiguous()
        t = None
        if kind == ops.CURR:
            t = head.t
        elif kind == ops.ADA:
            t = torch.cat([head.batch_mean.view(1), head.batch_std.view(1)]).float().to(feats.device)
        elif kind in (ops.ELASTIC_ARC, ops.ELASTIC_COS):
            t = torch.empty(N, device=feats.device).normal_(head.m, head.std).clamp_(head.m - head.std, head.m + head.std)
        elif kind == ops.VPL:
            t = torch.cat([head.mem.reshape(-1), head.life.reshape(-1)]).float().to(feats.device)
        out = ops.head_forward(ctx, x, w.detach().contiguous(), labels.contiguous(), state_t=t, lamb=lamb, want_logits=True,
                               elastic_plus=bool(getattr(head, "plus", False)) and kind in (ops.ELASTIC_ARC, ops.ELASTIC_COS))
        if kind == ops.ADA:
            head.batch_mean, head.batch_std = t[0:1].clone(), t[1:2].clone()
        elif kind == ops.VPL:
            cd = head.num_classes * D
            head.mem, head.life = t[:cd].view(head.num_classes, D), t[cd:]
        is_mag = kind == ops.MAG
        logits, loss_g = _StandaloneHead.apply(feats, w, out["logits"], out["loss_g"][0] if is_mag else None, ctx, x,
                                               labels.contiguous(), t)
        one_hot = torch.zeros_like(out["cos_s"]).scatter_(1, labels.view(-1, 1), 1.0)
        return [out["cos_s"], logits], out["norms"].view(-1, 1), (loss_g if is_mag else 0), one_hot

    @staticmethod
    def forward(ctx, feats, w, logits, loss_g, hctx, x, labels, t):
        ctx.hctx, ctx.x, ctx.labels, ctx.t = hctx, x, labels, t
        ctx.save_for_backward(w)
        ctx.set_materialize_grads(False)
        return logits.view_as(logits), (None if loss_g is None else loss_g.view_as(loss_g))

    @staticmethod
    def backward(ctx, dlogits, dloss_g):
        (w,) = ctx.saved_tensors
        if dlogits is None:
            dlogits = torch.zeros(ctx.hctx.desc.N, ctx.hctx.desc.C, device=ctx.x.device)
        if ctx.hctx.desc.kind == ops.MAG:
            ctx.hctx.desc.lamb = 0.0 if dloss_g is None else float(dloss_g)
        dx, dw = ops.head_backward_dlogits(ctx.hctx, ctx.x, w.detach().contiguous(), ctx.labels,
                                           dlogits.contiguous().float(), state_t=ctx.t)
        return dx, dw, None, None, None, None, None, None


def _net(name, kind, attr, make_head):
    class Net(NativeFaceNet):
        head_attr = attr

        def __init__(self, num_classes, backbone):
            head = make_head(num_classes)
            super().__init__(head, get_backbone(backbone), COMPUTE_DTYPE)
            self.kind = head.kind if kind is None else kind      # (MV-Softmax: 'am' / 'arc' is a constructor argument)
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


class MV_SoftmaxNet(_net(f"mv_softmax_{MARGIN_TYPE_mv}", None, "mv_head",
                         lambda c: MV_Softmax(FEATURE_DIM, c, margin=M_mv, mv_weight=WEIGHT_mv, s=S_mv, margin_type=MARGIN_TYPE_mv))):
    """criterion.py:463-489"""


class AdaFaceNet(_net("adaface", ops.ADA, "adaface", lambda c: AdaFace(FEATURE_DIM, c, m=M_ada, h=H_ada, s=S_ada, t_alpha=T_ALPHA_ada))):
    """criterion.py:920-949"""


class ElasticCosFaceNet(_net("elastic_cosface", ops.ELASTIC_COS, "head",
                             lambda c: ElasticCosFace(FEATURE_DIM, c, s=S_elastic_cos, m=M_elastic_cos, std=STD_elastic_cos,
                                                      plus=PLUS_elastic_cos))):
    """criterion.py:1032-1052"""


class ElasticArcFaceNet(_net("elastic_arcface", ops.ELASTIC_ARC, "head",
                             lambda c: ElasticArcFace(FEATURE_DIM, c, s=S_elastic_arc, m=M_elastic_arc, std=STD_elastic_arc,
                                                      plus=PLUS_elastic_arc))):
    """criterion.py:1156-1176"""


class MagFaceNet(_net("magface", ops.MAG, "magface",
                      lambda c: MagFace(FEATURE_DIM, c, s=S_mag, easy_margin=EASY_MARGIN_mag, l_margin=L_MARGIN_mag,
                                        u_margin=U_MARGIN_mag, l_a=L_A_mag, u_a=U_A_mag))):
    """criterion.py:1303-1329"""


class VPLArcFaceNet(_net("vpl_arcface", ops.VPL, "vpl_head",
                         lambda c: VPLArcFace(FEATURE_DIM, c, s=S_vpl, m=M_vpl, easy_margin=EASY_MARGIN_vpl, lamda=LAMDA_vpl,
                                              delta=DELTA_vpl))):
    """criterion.py:764-793"""

    def change_training_mode(self, flag):
        self.vpl_head.change_training_mode(flag)
