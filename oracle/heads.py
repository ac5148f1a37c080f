"""Oracle: the four margin heads + softmax-CE + top-k, closed-form forward AND backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy, any float dtype
(float32 = "the reference CPU path"; float64 for derivations).

Follows the reference (paths relative to /root/reference/main_code):
  ArcFace        utils/criterion.py:260-301   (ctor :234-249)
  CosFace        utils/criterion.py:162-197   (ctor :141-154)
  SphereFace     utils/criterion.py:57-107    (ctor :17-49)
  CurricularFace utils/criterion.py:527-587   (ctor :496-519)
  CE             utils/model_utils.py:556,179 (nn.CrossEntropyLoss, mean)
  accuracy       utils/metrics.py:3-16
Pinned by tests/golden/heads_*.npz (generated from the reference import).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

ARC, COS, SPHERE, CURR = 0, 1, 2, 3
KIND_NAMES = {ARC: "arcface", COS: "cosface", SPHERE: "sphereface", CURR: "curricular"}

NORM_EPS = 1e-12  # F.normalize default eps


@dataclass
class HeadState:
    """Mutable per-head state the reference keeps outside autograd."""
    iter: int = 0            # SphereFace.iter        (criterion.py:33,58)
    lamb: float = 0.0        # SphereFace.lamb        (criterion.py:60)
    t: float = 0.0           # CurricularFace.t buffer (criterion.py:517,572)


@dataclass
class HeadHyper:
    kind: int
    s: float = 64.0
    m: float = 0.5
    momentum: float = 0.01   # CurricularFace only
    # SphereFace annealing constants (criterion.py:29-33)
    base: float = 1000.0
    gamma: float = 0.12
    power: float = 1.0
    lambda_min: float = 5.0

    @staticmethod
    def default(kind: int) -> "HeadHyper":
        # utils/config.py:16-37
        if kind == ARC:
            return HeadHyper(ARC, s=64.0, m=0.5)
        if kind == COS:
            return HeadHyper(COS, s=64.0, m=0.35)
        if kind == SPHERE:
            return HeadHyper(SPHERE, s=1.0, m=2)  # S_sphere is unused by the reference
        if kind == CURR:
            return HeadHyper(CURR, s=64.0, m=0.5, momentum=0.01)
        raise ValueError(kind)


def weight_is_cd(kind: int) -> bool:
    """True when the class weight is stored [C, D] (ArcFace/SphereFace 'weight'),
    False when stored [D, C] (CosFace/CurricularFace 'kernel').  SURVEY H7."""
    return kind in (ARC, SPHERE)


@dataclass
class HeadOut:
    cos_s: np.ndarray        # first element of the reference's output list
    logits: np.ndarray       # second element (goes into CE)
    norms: np.ndarray        # [N,1]
    loss: float
    top1: int                # count of rows whose target is the arg-max of cos_s
    top5: int                # count of rows whose target is within the top 5
    dx: np.ndarray
    dw: np.ndarray           # same layout as the weight passed in
    lse: np.ndarray
    extra: dict = field(default_factory=dict)


def _normalize_rows(a, dt):
    n = np.sqrt((a.astype(dt) ** 2).sum(axis=1, keepdims=True))
    return a / np.maximum(n, dt(NORM_EPS)), n


def head_forward_backward(kind, x, w, labels, hyper: HeadHyper, state: HeadState,
                          dtype=np.float32, need_grad=True) -> HeadOut:
    """One training-mode forward of the head + mean CE + its analytic backward.

    x [N,D]; w [C,D] (ARC/SPHERE) or [D,C] (COS/CURR); labels [N] int.
    Mutates `state` exactly as the reference forward does.
    """
    dt = np.dtype(dtype).type
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    labels = np.asarray(labels).astype(np.int64)
    N, D = x.shape
    wc = w if weight_is_cd(kind) else w.T          # [C,D] view
    C = wc.shape[0]
    rows = np.arange(N)

    xh, xnorm = _normalize_rows(x, dt)             # x^, ||x||   (criterion.py:263,173,65,542)
    wh, wnorm = _normalize_rows(wc, dt)            # w^ per class (criterion.py:264,174,541)
    c_raw = (xh @ wh.T).astype(dtype)              # cosine GEMM (criterion.py:267,176,65,545)
    onehot = np.zeros((N, C), dtype=dtype)
    onehot[rows, labels] = 1

    extra = {}
    dnorm_coef = None
    if kind == ARC:
        cos_m, sin_m = dt(math.cos(hyper.m)), dt(math.sin(hyper.m))
        th, mm = dt(math.cos(math.pi - hyper.m)), dt(math.sin(math.pi - hyper.m) * hyper.m)
        c = c_raw                                                  # no clamp (criterion.py:280)
        q = dt(1.0) - c * c
        sine = np.sqrt(np.clip(q, dt(1e-9), dt(1.0)))              # :281
        phi = c * cos_m - sine * sin_m                             # :282
        phi = np.where(c > th, phi, c - mm)                        # :287 (easy_margin=False)
        z = (onehot * phi + (dt(1.0) - onehot) * c) * dt(hyper.s)  # :294-295
        cos_s = c * dt(hyper.s)
        inside = (q >= dt(1e-9)) & (q <= dt(1.0))                  # clamp passes grad inclusively
        dphi = np.where(c > th, cos_m + np.where(inside, sin_m * c / sine, dt(0)), dt(1.0))
        dzdc = dt(hyper.s) * np.where(onehot > 0, dphi, dt(1.0))
        pass_clamp = np.ones_like(c, dtype=bool)
    elif kind == COS:
        eps = dt(1e-4)
        lo, hi = dt(-1) + eps, dt(1) - eps
        c = np.clip(c_raw, lo, hi)                                 # :177
        z = (c - onehot * dt(hyper.m)) * dt(hyper.s)               # :186-189
        cos_s = c * dt(hyper.s)
        pass_clamp = (c_raw >= lo) & (c_raw <= hi)
        dzdc = np.full_like(c, dt(hyper.s))
    elif kind == SPHERE:
        state.iter += 1                                            # :58
        state.lamb = max(hyper.lambda_min,
                         hyper.base * (1 + hyper.gamma * state.iter) ** (-hyper.power))  # :60
        lam = dt(state.lamb)
        m = int(hyper.m)
        assert m == 2, "only m=2 (config.py:17) is on the hot path"
        c = np.clip(c_raw, dt(-1), dt(1))                          # :81
        cos_m_theta = dt(2) * c * c - dt(1)                        # mlambda[2], :44
        theta = np.arccos(c)                                       # :88 (detached)
        k = np.floor(dt(m) * theta / dt(math.pi))                  # :89
        sign = np.where(np.mod(k, 2) == 0, dt(1), dt(-1))
        phi = sign * cos_m_theta - dt(2) * k                       # :92
        u = onehot * (phi - c) / (dt(1) + lam) + c                 # :104
        z = u * xnorm                                              # :105
        cos_s = c * xnorm
        pass_clamp = (c_raw >= dt(-1)) & (c_raw <= dt(1))
        dudc = dt(1) + onehot * (sign * dt(4) * c - dt(1)) / (dt(1) + lam)
        dzdc = dudc * xnorm
        dnorm_coef = u                                             # dz/d||x|| = u
        extra.update(lamb=state.lamb, iter=state.iter)
    elif kind == CURR:
        cos_m, sin_m = dt(math.cos(hyper.m)), dt(math.sin(hyper.m))
        th, mm = dt(math.cos(math.pi - hyper.m)), dt(math.sin(math.pi - hyper.m) * hyper.m)
        c = np.clip(c_raw, dt(-1), dt(1))                          # :546
        ty = c[rows, labels][:, None]                              # :552
        sin_t = np.sqrt(dt(1.0) - ty * ty)                         # :555 (no eps)
        cm = ty * cos_m - sin_t * sin_m                            # :556
        mask = c > cm                                              # :559
        final_t = np.where(ty > th, cm, ty - mm)                   # :562-566
        t_new = dt(ty.mean(dtype=dtype)) * dt(hyper.momentum) + dt(1 - hyper.momentum) * dt(state.t)  # :572
        state.t = float(t_new)
        t = dt(t_new)
        zc = np.where(mask, c * (t + c), c)                        # :575
        zc[rows, labels] = final_t[:, 0]                           # :578
        z = zc * dt(hyper.s)                                       # :581
        cos_s = c * dt(hyper.s)
        pass_clamp = (c_raw >= dt(-1)) & (c_raw <= dt(1))
        dnon = np.where(mask, t + dt(2) * c, dt(1))
        dtar = np.where(ty > th, cos_m + sin_m * ty / sin_t, dt(1))
        dzdc = dt(hyper.s) * np.where(onehot > 0, np.broadcast_to(dtar, c.shape), dnon)
        extra.update(t=state.t)
    else:
        raise ValueError(kind)

    # ---- CE (mean) : nn.CrossEntropyLoss, model_utils.py:556,179
    zmax = z.max(axis=1, keepdims=True)
    e = np.exp(z - zmax)
    se = e.sum(axis=1, keepdims=True, dtype=dtype)
    lse = (zmax + np.log(se))[:, 0]
    loss = float((lse - z[rows, labels]).mean(dtype=np.float64))

    # ---- top-1 / top-5 on the PRE-margin scaled cosines (metrics.py:3-16, model_utils.py:182)
    ty_s = cos_s[rows, labels][:, None]
    rank = (cos_s > ty_s).sum(axis=1)
    top1 = int((rank < 1).sum())
    top5 = int((rank < 5).sum())

    dx = dw = None
    if need_grad:
        g = (e / se - onehot) / dt(N)                              # dL/dz
        dc = g * dzdc * pass_clamp                                 # dL/dc_raw
        dxh = dc @ wh                                              # [N,D]
        dwh = dc.T @ xh                                            # [C,D]
        dx = (dxh - xh * (xh * dxh).sum(axis=1, keepdims=True)) / np.maximum(xnorm, dt(NORM_EPS))
        dwc = (dwh - wh * (wh * dwh).sum(axis=1, keepdims=True)) / np.maximum(wnorm, dt(NORM_EPS))
        if dnorm_coef is not None:                                 # SphereFace: grad through ||x||
            dn = (g * dnorm_coef).sum(axis=1, keepdims=True)
            dx = dx + dn * xh
        dw = dwc if weight_is_cd(kind) else dwc.T
        dx = dx.astype(dtype)
        dw = np.ascontiguousarray(dw.astype(dtype))

    return HeadOut(cos_s=cos_s.astype(dtype), logits=z.astype(dtype), norms=xnorm.astype(dtype),
                   loss=loss, top1=top1, top5=top5, dx=dx, dw=dw, lse=lse, extra=extra)


def accuracy_topk(output, target, topk=(1,)):
    """metrics.py:3-16 -- percentages; ties resolved by 'strictly greater' count."""
    output = np.asarray(output)
    target = np.asarray(target).astype(np.int64)
    n = target.shape[0]
    ty = output[np.arange(n), target][:, None]
    rank = (output > ty).sum(axis=1)
    return [100.0 * float((rank < k).sum()) / n for k in topk]


def custom_step_lr(base_lr, epochs, steps=(20, 40, 60), ratio=0.1):
    """schedulers.py:3-14,20 -- lr in effect during epoch index e (0-based count of
    scheduler.step() calls), chained multiplication when last_epoch hits a step."""
    lrs, lr = [], base_lr
    for e in range(epochs):
        if e in set(steps):      # last_epoch == e after e calls to step()
            lr = lr * ratio
        lrs.append(lr)
    return lrs


def sgd_step(p, g, buf, lr, momentum=0.9, weight_decay=5e-4, first=False):
    """torch.optim.SGD step (model_utils.py:557): d = g + wd*p; buf = mu*buf + d
    (buf = d on the first step); p -= lr*buf.  Returns (p, buf)."""
    d = g + weight_decay * p
    buf = d.copy() if first else momentum * buf + d
    return p - lr * buf, buf
