"""Oracle: the four margin heads + softmax-CE + top-k, closed-form forward AND backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy, any float dtype
(float32 = "the reference CPU path"; float64 for derivations).

Follows the reference (paths relative to /root/reference/main_code):
  ArcFace        utils/criterion.py:260-301   (ctor :234-249)
  CosFace        utils/criterion.py:162-197   (ctor :141-154)
  SphereFace     utils/criterion.py:57-107    (ctor :17-49)
  CurricularFace utils/criterion.py:527-587   (ctor :496-519)
  MV_Softmax     utils/criterion.py:388-450   (ctor :334-377; margin types 'am' and 'arc')
  AdaFace        utils/criterion.py:848-907   (ctor :802-841)
  ElasticArcFace utils/criterion.py:1089-1145 (ctor :1061-1083; plus=False and plus=True)
  ElasticCosFace utils/criterion.py:982-1021  (ctor :955-976; plus=False and plus=True)
  MagFace        utils/criterion.py:1241-1291 (ctor :1185-1222)
  VPLArcFace     utils/criterion.py:686-752   (ctor :626-674)
  CE             utils/model_utils.py:556,179 (nn.CrossEntropyLoss, mean)
  accuracy       utils/metrics.py:3-16
Pinned by tests/golden/heads_*.npz (generated from the reference import: make_golden.py, make_golden_heads2.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

ARC, COS, SPHERE, CURR = 0, 1, 2, 3
MV_AM, MV_ARC, ADA, ELASTIC_ARC, ELASTIC_COS, MAG, VPL = 4, 5, 6, 7, 8, 9, 10
KIND_NAMES = {ARC: "arcface", COS: "cosface", SPHERE: "sphereface", CURR: "curricular", MV_AM: "mv_am", MV_ARC: "mv_arc",
              ADA: "adaface", ELASTIC_ARC: "elastic_arc", ELASTIC_COS: "elastic_cos", MAG: "magface", VPL: "vpl_arcface"}

NORM_EPS = 1e-12  # F.normalize default eps


@dataclass
class HeadState:
    """Mutable per-head state the reference keeps outside autograd."""
    iter: int = 0            # SphereFace.iter        (criterion.py:33,58)
    lamb: float = 0.0        # SphereFace.lamb        (criterion.py:60)
    t: float = 0.0           # CurricularFace.t buffer (criterion.py:517,572)
    batch_mean: float = 20.0   # AdaFace buffers (criterion.py:838-839)
    batch_std: float = 100.0
    mem: object = None         # VPLArcFace buffers (criterion.py:660-661): [C,D] and [C]; created on first use
    life: object = None


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
    mv_weight: float = 1.12  # MV_Softmax (config.py:30)
    h: float = 0.333         # AdaFace (config.py:49-50)
    t_alpha: float = 0.99
    l_margin: float = 0.45   # MagFace (config.py:66-70)
    u_margin: float = 0.8
    l_a: float = 10.0
    u_a: float = 110.0
    easy_margin: bool = False
    lamda: float = 0.15      # VPLArcFace (config.py:43-44)
    delta: float = 100.0
    memory_on: bool = True   # VPLArcFace.norm_training_flag (criterion.py:671)
    plus: bool = False       # Elastic heads: rank-matched margins (criterion.py:1006-1011, 1117-1122)

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
        if kind in (MV_AM, MV_ARC):                     # config.py:29-32
            return HeadHyper(kind, s=32.0, m=0.35, mv_weight=1.12)
        if kind == ADA:                                  # config.py:47-50
            return HeadHyper(ADA, s=64.0, m=0.4, h=0.333, t_alpha=0.99)
        if kind == ELASTIC_ARC:                          # config.py:53-56 (the margin itself is sampled per row)
            return HeadHyper(ELASTIC_ARC, s=64.0, m=0.5)
        if kind == ELASTIC_COS:                          # config.py:59-62
            return HeadHyper(ELASTIC_COS, s=64.0, m=0.35)
        if kind == MAG:                                  # config.py:65-70
            return HeadHyper(MAG, s=64.0, m=0.0)
        if kind == VPL:                                  # config.py:40-44
            return HeadHyper(VPL, s=64.0, m=0.5, easy_margin=False, lamda=0.15, delta=100.0)
        raise ValueError(kind)


def weight_is_cd(kind: int) -> bool:
    """True when the class weight is stored [C, D] (ArcFace/SphereFace 'weight'),
    False when stored [D, C] (CosFace/CurricularFace 'kernel').  SURVEY H7."""
    return kind in (ARC, SPHERE, MV_AM, MV_ARC, VPL)


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
                          dtype=np.float32, need_grad=True, row_margin=None, lambda_g=0.0) -> HeadOut:
    """One training-mode forward of the head + mean CE + its analytic backward.

    x [N,D]; w [C,D] (ARC/SPHERE/MV) or [D,C] (the others); labels [N] int.
    Mutates `state` exactly as the reference forward does.
    row_margin [N]: the elastic heads' sampled (and clamped) margins -- the reference draws them with torch.normal
    inside forward (criterion.py:1002-1004, 1113-1115); here they are an input.
    lambda_g: MagFace only; dx is the gradient of  CE + lambda_g * loss_g  (model_utils.py:180); `loss` stays the CE.
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
        on = (c > dt(0)) if hyper.easy_margin else (c > th)
        phi = np.where(on, phi, c if hyper.easy_margin else c - mm)  # :284-287
        z = (onehot * phi + (dt(1.0) - onehot) * c) * dt(hyper.s)  # :294-295
        cos_s = c * dt(hyper.s)
        inside = (q >= dt(1e-9)) & (q <= dt(1.0))                  # clamp passes grad inclusively
        dphi = np.where(on, cos_m + np.where(inside, sin_m * c / sine, dt(0)), dt(1.0))
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
        assert 1 <= m <= 5, "the Chebyshev table (criterion.py:40-47) covers m = 0..5"
        c = np.clip(c_raw, dt(-1), dt(1))                          # :81
        c2 = c * c
        cheb = {1: (c, np.ones_like(c)), 2: (dt(2) * c2 - dt(1), dt(4) * c),
                3: ((dt(4) * c2 - dt(3)) * c, dt(12) * c2 - dt(3)),
                4: ((dt(8) * c2 - dt(8)) * c2 + dt(1), (dt(32) * c2 - dt(16)) * c),
                5: (((dt(16) * c2 - dt(20)) * c2 + dt(5)) * c, (dt(80) * c2 - dt(60)) * c2 + dt(5))}
        cos_m_theta, dcheb = cheb[m]                               # mlambda[m], :40-47, and its derivative
        theta = np.arccos(c)                                       # :88 (detached)
        k = np.floor(dt(m) * theta / dt(math.pi))                  # :89
        sign = np.where(np.mod(k, 2) == 0, dt(1), dt(-1))
        phi = sign * cos_m_theta - dt(2) * k                       # :92
        u = onehot * (phi - c) / (dt(1) + lam) + c                 # :104
        z = u * xnorm                                              # :105
        cos_s = c * xnorm
        pass_clamp = (c_raw >= dt(-1)) & (c_raw <= dt(1))
        dudc = dt(1) + onehot * (sign * dcheb - dt(1)) / (dt(1) + lam)
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
    elif kind in (MV_AM, MV_ARC):
        lo, hi = dt(-1) + dt(1e-7), dt(1) - dt(1e-7)
        c = np.clip(c_raw, lo, hi)                                 # :414
        ty = c[rows, labels][:, None]                              # :418
        if kind == MV_AM:
            final_t = np.where(ty > dt(hyper.m), ty - dt(hyper.m), ty)      # :422-424
            thr = ty - dt(hyper.m)
            dtar = np.ones_like(ty)
        else:
            cos_m, sin_m = dt(math.cos(hyper.m)), dt(math.sin(hyper.m))
            sin_t = np.sqrt(dt(1.0) - ty * ty + dt(1e-9))          # :428
            thr = ty * cos_m - sin_t * sin_m                       # :429
            final_t = np.where(ty > 0, thr, ty)                    # :430
            dtar = np.where(ty > 0, cos_m + sin_m * ty / sin_t, dt(1))
        mask = c > thr                                             # :425 / :431
        mvw = dt(hyper.mv_weight)
        zc = np.where(mask, mvw * c + (mvw - dt(1.0)), c)          # :434-436
        zc[rows, labels] = final_t[:, 0]                           # :440
        z = zc * dt(hyper.s)
        cos_s = c * dt(hyper.s)
        pass_clamp = (c_raw >= lo) & (c_raw <= hi)
        dnon = np.where(mask, mvw, dt(1))
        dzdc = dt(hyper.s) * np.where(onehot > 0, np.broadcast_to(dtar, c.shape), dnon)
    elif kind == ADA:
        eps = dt(1e-3)
        lo, hi = dt(-1) + eps, dt(1) - eps
        c = np.clip(c_raw, lo, hi)                                 # :866
        safe = np.clip(xnorm, dt(0.001), dt(100))                  # :870
        mean, std = safe.mean(dtype=dtype), safe.std(ddof=1, dtype=dtype)     # :873-874 (unbiased)
        ta = dt(hyper.t_alpha)
        state.batch_mean = float(mean * ta + (dt(1) - ta) * dt(state.batch_mean))   # :875-876
        state.batch_std = float(std * ta + (dt(1) - ta) * dt(state.batch_std))
        ms = (safe - dt(state.batch_mean)) / (dt(state.batch_std) + eps)     # :878
        ms = np.clip(ms * dt(hyper.h), dt(-1), dt(1))              # :879
        theta = np.arccos(c)                                       # :887
        raw = theta + onehot * (dt(hyper.m) * ms * dt(-1))         # :888-889
        tlo, thi = eps, dt(math.pi) - eps
        theta_m = np.clip(raw, tlo, thi)
        zc = np.cos(theta_m) - onehot * (dt(hyper.m) + dt(hyper.m) * ms)    # :890-895
        z = zc * dt(hyper.s)
        cos_s = c * dt(hyper.s)
        pass_clamp = (c_raw >= lo) & (c_raw <= hi)
        inside = (raw >= tlo) & (raw <= thi)
        dzdc = dt(hyper.s) * np.where(inside, np.sin(theta_m) / np.sqrt(dt(1) - c * c), dt(0))
        extra.update(batch_mean=state.batch_mean, batch_std=state.batch_std, row_param=ms[:, 0].copy())
    elif kind in (ELASTIC_ARC, ELASTIC_COS):
        lo, hi = dt(-1) + dt(1e-7), dt(1) - dt(1e-7)
        c = np.clip(c_raw, lo, hi)                                 # :997 / :1108
        mrow = np.asarray(row_margin, dtype=dtype).reshape(N, 1)
        if hyper.plus:                                             # :1006-1011 / :1117-1122, indexing as written there:
            rank = np.argsort(-c[rows, labels], kind="stable")     #   _, rank = sort(target_cos, descending)
            mrow = np.sort(mrow[:, 0])[rank].reshape(N, 1)         #   margin = sort(margin)[rank]
        if kind == ELASTIC_COS:
            zc = c - onehot * mrow                                 # :1013
            dzdc = np.full_like(c, dt(hyper.s))
        else:
            ty = c[rows, labels][:, None]
            raw = np.arccos(ty) + mrow                             # :1126-1127
            theta_m = np.clip(raw, dt(0), dt(math.pi))             # :1128
            zc = c.copy()
            zc[rows, labels] = np.cos(theta_m)[:, 0]               # :1129-1132
            inside = (raw >= dt(0)) & (raw <= dt(math.pi))
            dtar = np.where(inside, np.sin(theta_m) / np.sqrt(dt(1) - ty * ty), dt(0))
            dzdc = dt(hyper.s) * np.where(onehot > 0, np.broadcast_to(dtar, c.shape), dt(1))
        z = zc * dt(hyper.s)
        cos_s = c * dt(hyper.s)
        pass_clamp = (c_raw >= lo) & (c_raw <= hi)
        extra.update(row_param=mrow[:, 0].copy())
    elif kind == VPL:
        lo, hi = dt(-1) + dt(1e-7), dt(1) - dt(1e-7)
        vpl_al = None
        blend = c_raw                                              # cosine_weight (:694-695)
        if hyper.memory_on:
            if state.mem is None:
                state.mem, state.life = np.zeros((C, D), dtype=dtype), np.zeros(C, dtype=dtype)
            for cls in np.unique(labels):                          # :703-709 (raw features, per-class mean)
                state.mem[cls] = x[labels == cls].mean(axis=0, dtype=dtype)
                state.life[cls] = hyper.delta
            state.life = state.life - 1                            # :712
            vpl_al = ((state.life > 0).astype(dtype) * dt(hyper.lamda))[None, :]       # active * lamda, [1,C]
            mh, _ = _normalize_rows(state.mem.astype(dtype), dt)   # :716
            vpl_mh = mh
            c_mem = (xh @ mh.T).astype(dtype)                      # :717
            cos1 = (dt(1) - vpl_al) * c_raw + vpl_al * c_mem       # :720
            cos2 = (dt(1) - vpl_al) * c_raw + vpl_al * dt(1.0)     # :721
            blend = onehot * cos2 + (dt(1.0) - onehot) * cos1      # :722
        c = np.clip(blend, lo, hi)                                 # :730
        cos_m, sin_m = dt(math.cos(hyper.m)), dt(math.sin(hyper.m))
        th, mm = dt(math.cos(math.pi - hyper.m)), dt(math.sin(math.pi - hyper.m) * hyper.m)
        sine = np.sqrt(dt(1.0) - c * c + dt(1e-9))                 # :734
        phi = c * cos_m - sine * sin_m
        if hyper.easy_margin:
            on, off = c > 0, c                                     # :738
        else:
            on, off = c > th, c - mm                               # :740
        z = (onehot * np.where(on, phi, off) + (dt(1.0) - onehot) * c) * dt(hyper.s)   # :743-744
        cos_s = c * dt(hyper.s)
        pass_clamp = (blend >= lo) & (blend <= hi)
        dzdc = dt(hyper.s) * np.where(onehot > 0, np.where(on, cos_m + sin_m * c / sine, dt(1)), dt(1))
    elif kind == MAG:
        lo, hi = dt(-1) + dt(1e-7), dt(1) - dt(1e-7)
        la, ua = dt(hyper.l_a), dt(hyper.u_a)
        xn = np.clip(xnorm, la, ua)                                # :1246
        loss_g = float((dt(1) / (ua ** 2) * xn + dt(1) / xn).mean(dtype=np.float64))   # :1235-1239
        c = np.clip(c_raw, lo, hi)                                 # :1261
        slope = dt((hyper.u_margin - hyper.l_margin) / (hyper.u_a - hyper.l_a))
        am = slope * (xn - la) + dt(hyper.l_margin)                # :1229-1233
        cos_m, sin_m = np.cos(am), np.sin(am)
        ty = c[rows, labels][:, None]
        sin_t = np.sqrt(dt(1.0) - ty * ty + dt(1e-9))              # :1270
        ctm = ty * cos_m - sin_t * sin_m                           # :1271
        d_c = cos_m + sin_m * ty / sin_t
        d_m = -ty * sin_m - sin_t * cos_m
        if hyper.easy_margin:
            on = ty > 0                                            # :1274
            off_z, off_dm = ty, np.zeros_like(ty)
        else:
            mm = np.sin(dt(math.pi) - am) * am                     # :1277
            on = ty > np.cos(dt(math.pi) - am)                     # :1278-1279
            off_z = ty - mm
            off_dm = np.cos(dt(math.pi) - am) * am - np.sin(dt(math.pi) - am)
        final_t = np.where(on, ctm, off_z)
        zc = c.copy()
        zc[rows, labels] = final_t[:, 0]                           # :1286
        z = zc * dt(hyper.s)                                       # :1287
        cos_s = c * dt(hyper.s)
        pass_clamp = (c_raw >= lo) & (c_raw <= hi)
        dtar = np.where(on, d_c, dt(1))
        dzdc = dt(hyper.s) * np.where(onehot > 0, np.broadcast_to(dtar, c.shape), dt(1))
        norm_pass = ((xnorm >= la) & (xnorm <= ua)).astype(dtype)
        # d z_target / d||x|| through the adaptive margin, and d loss_g / d||x||
        mag_dz_dnorm = dt(hyper.s) * np.where(on, d_m, off_dm) * slope * norm_pass
        mag_dg_dnorm = (dt(1) / (ua ** 2) - dt(1) / (xn * xn)) * norm_pass / dt(N)
        extra.update(loss_g=loss_g, row_param=am[:, 0].copy(), x_norm=xn.astype(dtype))
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
        if kind == VPL and hyper.memory_on:                        # through the blend: weight path and memory path
            dc_mem = dc * vpl_al * (dt(1.0) - onehot)
            dc = dc * (dt(1) - vpl_al)
        dxh = dc @ wh                                              # [N,D]
        if kind == VPL and hyper.memory_on:
            dxh = dxh + dc_mem @ vpl_mh
        dwh = dc.T @ xh                                            # [C,D]
        dx = (dxh - xh * (xh * dxh).sum(axis=1, keepdims=True)) / np.maximum(xnorm, dt(NORM_EPS))
        dwc = (dwh - wh * (wh * dwh).sum(axis=1, keepdims=True)) / np.maximum(wnorm, dt(NORM_EPS))
        if dnorm_coef is not None:                                 # SphereFace: grad through ||x||
            dn = (g * dnorm_coef).sum(axis=1, keepdims=True)
            dx = dx + dn * xh
        if kind == MAG:                                            # grad through ||x||: adaptive margin + loss_g
            dn = g[rows, labels][:, None] * mag_dz_dnorm + dt(lambda_g) * mag_dg_dnorm
            dx = dx + dn * xh
        dw = dwc if weight_is_cd(kind) else dwc.T
        dx = dx.astype(dtype)
        dw = np.ascontiguousarray(dw.astype(dtype))

    return HeadOut(cos_s=cos_s.astype(dtype), logits=z.astype(dtype),
                   norms=(extra["x_norm"] if kind == MAG else xnorm).astype(dtype),
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
