"""Thin tensor-level wrappers over the C ABI.  torch is used for device memory and
streams only.  Every function requires HIP tensors and a loaded libfrx.so."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import FrxError, HeadDesc, check

ARC, COS, SPHERE, CURR = 0, 1, 2, 3
MV_AM, MV_ARC, ADA, ELASTIC_ARC, ELASTIC_COS, MAG, VPL = 4, 5, 6, 7, 8, 9, 10       # include/frx.h frx_head_kind
W_CD_KINDS = (ARC, SPHERE, MV_AM, MV_ARC, VPL)       # heads whose parameter is `weight` [C, D]; the others hold `kernel` [D, C]

# Optional per-launch timing (bench.py roofline leg): when PROFILER is a list, the GEMM-class entry
# points bracket their launch with events on the current stream and append
# (kernel label, algorithmic FLOPs, start event, end event).
PROFILER = None


def _timed(label, flops, dev_tensor, fn, nbytes=0):
    if PROFILER is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(dev_tensor.device)
    e0.record(st)
    r = fn()
    e1.record(st)
    PROFILER.append((label() if callable(label) else label, flops, e0, e1, nbytes))
    return r


_MODE = {0: "fwd", 1: "dgrad", 2: "stem", 3: "fwd3x3patch", 4: "dgrad3x3patch"}
_PRO = {0: "none", 1: "bn_relu", 2: "bn_bwd", 3: "merge"}
_EPI = {0: "plain", 1: "stats", 2: "bnbwd_stats", 3: "fc", 4: "bnbwd_stats_maskout"}


def igemm_class(dtype_name, f):
    """label of ONE k_igemm behaviour from frx_last_conv_launch's fields (the same string scripts/traffic_summary.py
    derives from the kernel symbol): tile, gather mode, prologue, epilogue, staging"""
    bm, bn, waves, kc, ns, mode, pro, epi, add, persist, spec = f[:11]
    stage = "patch" if mode in (3, 4) else (f"dma{ns}" if ns else "ring")
    if spec == 3:       # csrc/pw_stream.hip: a wave streams 16 pixels at a time, operand fragments straight from global memory, weights in LDS
        return f"k_pw_stream<{dtype_name},16x{bn}x{waves}w,{_MODE[mode]},pro={_PRO[pro]},epi={_EPI[epi]},stream>"
    if spec == 2:       # csrc/pw_rows.hip: a row block's operand resident in LDS, persistent blocks over (row block, column tile) items
        return f"k_pw_rows<{dtype_name},{bm}x{bn}x{waves}w,{_MODE[mode]},pro={_PRO[pro]},epi={_EPI[epi]}{'+add' if add else ''},rows>"
    return (f"k_igemm<{dtype_name},{bm}x{bn}x{waves}w,kc{kc},{_MODE[mode]},pro={_PRO[pro]},epi={_EPI[epi]}{'+add' if add else ''},"
            f"{stage}{',persist' if persist else ''}{',stagewaves' if spec else ''}>")


def _igemm_label(dtype):
    def get():
        f = (C.c_int * 12)()
        check(_lib.lib().frx_last_conv_launch(f), "frx_last_conv_launch")
        return igemm_class(_dt_name(dtype), list(f))
    return get


def _igemm_tile(d, dgrad=False):
    """block tile of the launch, for the profiling labels (frx_conv_tile: the library's own choice)"""
    bm, bn = C.c_int(0), C.c_int(0)
    check(_lib.lib().frx_conv_tile(C.byref(d), int(dgrad), C.byref(bm), C.byref(bn)), "frx_conv_tile")
    return bm.value, bn.value


def conv_patch_mode(d, dgrad=False):
    """frx_conv_patch_mode: the row tile (128 / 64) of the patch-mode 3x3 kernel if this layer's geometry takes it (given a
    prologue and no addend), else 0"""
    r = _lib.lib().frx_conv_patch_mode(C.byref(d), int(dgrad))
    if r < 0:
        raise FrxError("conv descriptor rejected: " + _lib.lib().frx_last_error().decode())
    return r


def _dt_name(dt):
    return "bf16" if dt == 1 else "f32"


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise FrxError("frx ops need tensors on the HIP device (there is no CPU path)")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _pv(t):
    """device address for a ctypes struct field (None -> NULL)"""
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if t.dtype != dtype or not t.is_contiguous():
        raise FrxError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")


class HeadContext:
    """Descriptor + workspace of one head instance (one per model; reused every step)."""

    def __init__(self, kind, N, D, C_, s, m, momentum=0.01, device=None, p=(0.0, 0.0, 0.0, 0.0), flags=0, lambda_g=0.0,
                 class_offset=None):
        """p / flags: per-kind parameters of frx_head_desc (MV: mv_weight; ADA: h, t_alpha; MAG: l_margin, u_margin,
        l_a, u_a and flags bit 0 = easy_margin; VPL: lamda, delta and flags bit 0 = easy_margin, bit 1 = memory in use).
        lambda_g: MagFace's loss_g weight for the fused backward.
        class_offset: not None = class-sharded mode: this context is the column shard [class_offset, class_offset + C_)
        of a wider head and N counts the rows of the gathered batch (head_shard_* below)."""
        p = tuple(float(v) for v in p) + (0.0,) * (4 - len(p))
        if class_offset is not None:
            flags = int(flags) | 8
        self.desc = HeadDesc(kind, N, D, C_, s, m, momentum, float(lambda_g) if kind == MAG else 0.0,
                             (C.c_float * 4)(*p), int(flags), int(class_offset or 0))
        nbytes = _lib.lib().frx_head_workspace_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise FrxError("head descriptor rejected: " + _lib.lib().frx_last_error().decode())
        self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.nbytes = nbytes

    @property
    def shape(self):
        return self.desc.N, self.desc.D, self.desc.C


def head_forward_cos(ctx: HeadContext, x, w, labels, state_t=None, ty_sum=None):
    """Phase 1 of the head forward: norms, the cosine GEMM into the workspace, (VPL: class memory + blend) and the
    per-row target cosines; their batch sum lands in `ty_sum` [1] (the only value a data-parallel CurricularFace has
    to exchange before phase 2, criterion.py:570-573)."""
    N, D, Cc = ctx.shape
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "w"); _chk(labels, torch.int64, "labels")
    if tuple(x.shape) != (N, D) or labels.numel() != N or w.numel() != D * Cc:
        raise FrxError(f"head_forward: shapes x{tuple(x.shape)} w{tuple(w.shape)} labels{tuple(labels.shape)} "
                       f"do not match the context (N={N}, D={D}, C={Cc})")
    dev, st = _dev(x), _stream(x)
    if ty_sum is None:
        ty_sum = torch.empty(1, device=x.device)
    L = _lib.lib()
    # (only CurricularFace's EMA consumes the sum: the other kinds do not ask the library for it -- on the fused forward of
    # ARC / COS / MV that saves a launch; their `ty_sum` tensor keeps whatever it held)
    check(L.frx_head_fwd_cos(dev, st, C.byref(ctx.desc), _p(x), _p(w), _p(labels), _p(ctx.ws), ctx.nbytes,
                             _p(ty_sum) if ctx.desc.kind == CURR else None), "frx_head_fwd_cos")
    if ctx.desc.kind == VPL:
        check(L.frx_head_vpl_prepare(dev, st, C.byref(ctx.desc), _p(x), _p(labels), _p(state_t), _p(ctx.ws), ctx.nbytes),
              "frx_head_vpl_prepare")
    return ty_sum


def head_target_cos(ctx: HeadContext, out):
    """out [N] <- clamped target cosines of the rows (valid after head_forward_cos)"""
    _chk(out, torch.float32, "out")
    check(_lib.lib().frx_head_target_cos(_dev(out), _stream(out), C.byref(ctx.desc), _p(ctx.ws), ctx.nbytes, _p(out)),
          "frx_head_target_cos")
    return out


def rank_matched_margins(ctx: HeadContext, margins, scratch):
    """Elastic heads, plus=True (criterion.py:1006-1011 / 1117-1122), in place on `margins` [N]:
    rank = argsort(target cosine, descending); margins <- sort(margins)[rank]   (the reference's own indexing)."""
    head_target_cos(ctx, scratch)
    rank = torch.sort(scratch, descending=True).indices
    margins.copy_(torch.sort(margins).values[rank])


def head_forward(ctx: HeadContext, x, w, labels, state_t=None, lamb=0.0, want_logits=False,
                 ty_allreduce=None, elastic_plus=False):
    """Both phases.  ty_allreduce: optional callable(tensor[1]) -> count, summing the target-cosine sum over
    data-parallel ranks between them (CurricularFace EMA).  elastic_plus: rank-match the margins in
    state_t to the target cosines between the phases."""
    tys = head_forward_cos(ctx, x, w, labels, state_t)
    if elastic_plus:
        rank_matched_margins(ctx, state_t, torch.empty_like(state_t))
    count = ctx.desc.N if ty_allreduce is None else ty_allreduce(tys)
    return head_forward_loss(ctx, labels, tys, count, state_t, lamb, want_logits)


def head_forward_loss(ctx: HeadContext, labels, ty_sum, count, state_t=None, lamb=0.0, want_logits=False):
    """Phase 2: state update from the (global) target-cosine sum over `count` samples, margin + CE + top-k row sweep.
    Returns dict(loss[1], topk[2] int32, norms[N], lse[N], cos_s, logits[, loss_g, row_param]).
    SphereFace: lamb=None reads this forward's annealing lambda from state_t[0] (frx_head_desc flags bit 2: a
    captured graph follows criterion.py:58-60); a number is passed by value."""
    N, D, Cc = ctx.shape
    if ctx.desc.kind == SPHERE:
        if lamb is None:
            ctx.desc.flags |= 4
        else:
            ctx.desc.flags &= ~4
            ctx.desc.lamb = float(lamb)
    dev, st = _dev(ty_sum), _stream(ty_sum)
    device = ty_sum.device
    o = dict(loss=torch.empty(1, device=device), topk=torch.empty(2, dtype=torch.int32, device=device),
             norms=torch.empty(N, device=device), lse=torch.empty(N, device=device), cos_s=None, logits=None)
    if want_logits:
        o["cos_s"] = torch.empty(N, Cc, device=device)
        o["logits"] = torch.empty(N, Cc, device=device)
    L = _lib.lib()
    check(L.frx_head_fwd_loss(dev, st, C.byref(ctx.desc), _p(labels), _p(state_t), _p(ty_sum), int(count),
                              _p(ctx.ws), ctx.nbytes, _p(o["cos_s"]), _p(o["logits"]), _p(o["norms"]),
                              _p(o["loss"]), _p(o["lse"]), _p(o["topk"])), "frx_head_fwd_loss")
    if ctx.desc.kind in (ADA, ELASTIC_ARC, ELASTIC_COS, MAG):
        o["loss_g"] = torch.empty(1, device=device)
        o["row_param"] = torch.empty(N, device=device)
        check(L.frx_head_aux(dev, st, C.byref(ctx.desc), _p(state_t), _p(ctx.ws), ctx.nbytes, _p(o["loss_g"]),
                             _p(o["row_param"])), "frx_head_aux")
    return o


# ---- class-sharded head: the four compute phases between the caller's collectives (include/frx.h)
def head_shard_cos(ctx: HeadContext, x, w, labels, ty_out):
    N, D, Cc = ctx.shape
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "w"); _chk(labels, torch.int64, "labels"); _chk(ty_out, torch.float32, "ty_out")
    if tuple(x.shape) != (N, D) or labels.numel() != N or w.numel() != D * Cc or ty_out.numel() != N:
        raise FrxError(f"head_shard_cos: shapes x{tuple(x.shape)} w{tuple(w.shape)} labels{tuple(labels.shape)} "
                       f"do not match the context (N={N}, D={D}, C={Cc})")
    check(_lib.lib().frx_head_shard_cos(_dev(x), _stream(x), C.byref(ctx.desc), _p(x), _p(w), _p(labels), _p(ctx.ws),
                                        ctx.nbytes, _p(ty_out)), "frx_head_shard_cos")
    return ty_out


def head_shard_rows(ctx: HeadContext, labels, ty_global, part, state_t=None):
    """part [3, N] <- (row max, sum exp(z - row max), local rank count) over this shard's columns"""
    N = ctx.shape[0]
    _chk(part, torch.float32, "part"); _chk(ty_global, torch.float32, "ty_global")
    if part.numel() != 3 * N or ty_global.numel() != N:
        raise FrxError("head_shard_rows: part must hold 3*N floats and ty_global N")
    check(_lib.lib().frx_head_shard_rows(_dev(part), _stream(part), C.byref(ctx.desc), _p(labels), _p(state_t), _p(ty_global),
                                         _p(ctx.ws), ctx.nbytes, _p(part)), "frx_head_shard_rows")
    return part


def head_shard_rescale(local_max, global_max, part_sum):
    check(_lib.lib().frx_head_shard_rescale(_dev(part_sum), _stream(part_sum), part_sum.numel(), _p(local_max), _p(global_max),
                                            _p(part_sum)), "frx_head_shard_rescale")


def head_shard_finish(ctx: HeadContext, global_max, global_sum, global_rank, state_t=None):
    N = ctx.shape[0]
    dev = global_max.device
    o = dict(loss=torch.empty(1, device=dev), topk=torch.empty(2, dtype=torch.int32, device=dev),
             norms=torch.empty(N, device=dev), lse=torch.empty(N, device=dev), cos_s=None, logits=None)
    check(_lib.lib().frx_head_shard_finish(_dev(global_max), _stream(global_max), C.byref(ctx.desc), _p(state_t), _p(global_max),
                                           _p(global_sum), _p(global_rank), _p(ctx.ws), ctx.nbytes, _p(o["norms"]), _p(o["loss"]),
                                           _p(o["lse"]), _p(o["topk"])), "frx_head_shard_finish")
    return o


def head_backward(ctx: HeadContext, x, w, labels, state_t=None, gout=None, dx=None, dw=None, accumulate_dw=False):
    N, D, Cc = ctx.shape
    dx = torch.empty_like(x) if dx is None else dx
    dw = torch.empty_like(w) if dw is None else dw
    _chk(dx, torch.float32, "dx"); _chk(dw, torch.float32, "dw")
    check(_lib.lib().frx_head_bwd(_dev(x), _stream(x), C.byref(ctx.desc), _p(x), _p(w), _p(labels), _p(state_t),
                                  _p(gout), _p(ctx.ws), ctx.nbytes, _p(dx), _p(dw), int(bool(accumulate_dw))),
          "frx_head_bwd")
    return dx, dw


def head_backward_dlogits(ctx: HeadContext, x, w, labels, dlogits, state_t=None, dx=None, dw=None, accumulate_dw=False):
    """Backward for an arbitrary dL/dlogits [N,C] (autograd-compatible path)."""
    N, D, Cc = ctx.shape
    _chk(dlogits, torch.float32, "dlogits")
    if tuple(dlogits.shape) != (N, Cc):
        raise FrxError(f"dlogits shape {tuple(dlogits.shape)} != ({N}, {Cc})")
    dx = torch.empty_like(x) if dx is None else dx
    dw = torch.empty_like(w) if dw is None else dw
    check(_lib.lib().frx_head_bwd_dlogits(_dev(x), _stream(x), C.byref(ctx.desc), _p(x), _p(w), _p(labels),
                                          _p(state_t), _p(dlogits), _p(ctx.ws), ctx.nbytes, _p(dx), _p(dw),
                                          int(bool(accumulate_dw))), "frx_head_bwd_dlogits")
    return dx, dw


def pair_cosine(f1, f2):
    _chk(f1, torch.float32, "f1"); _chk(f2, torch.float32, "f2")
    if f1.shape != f2.shape or f1.dim() != 2:
        raise FrxError("pair_cosine: f1/f2 must be [P,D] of equal shape")
    P, D = f1.shape
    out = torch.empty(P, device=f1.device)
    check(_lib.lib().frx_pair_cosine(_dev(f1), _stream(f1), _p(f1), _p(f2), P, D, _p(out)), "frx_pair_cosine")
    return out


def threshold_count(cos, same, thr):
    _chk(cos, torch.float32, "cos"); _chk(same, torch.int64, "same")
    cnt = torch.zeros(1, dtype=torch.int32, device=cos.device)
    check(_lib.lib().frx_threshold_count(_dev(cos), _stream(cos), _p(cos), _p(same), cos.numel(), float(thr), _p(cnt)),
          "frx_threshold_count")
    return cnt


# ------------------------------------------------------------------ backbone wrappers
from ._lib import ConvDesc  # noqa: E402

F32, BF16 = 0, 1
TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def conv_desc(dtype, N, Hi, Wi, Ci, Co, R, S, stride, pad, stem=False):
    Ho = (Hi + 2 * pad - R) // stride + 1
    Wo = (Wi + 2 * pad - S) // stride + 1
    return ConvDesc(dtype, N, Hi, Wi, Ci, Co, R, S, stride, pad, Ho, Wo, int(stem))


def stem_padded_dims(H, W):
    hp, wp = C.c_int(0), C.c_int(0)
    check(_lib.lib().frx_stem_padded_dims(H, W, C.byref(hp), C.byref(wp)), "frx_stem_padded_dims")
    return hp.value, wp.value


def conv_stat_rows(d):
    r = _lib.lib().frx_conv_stat_rows(C.byref(d))
    if r < 0:
        raise FrxError("conv descriptor rejected: " + _lib.lib().frx_last_error().decode())
    return r


def _esz(d):
    return 2 if d.dtype == 1 else 4


def conv_bytes(d, n_in=1, n_out=1, wbytes=None):
    """algorithmic HBM bytes of one conv-class launch: every operand tensor once (a 3x3 conv still reads its
    input once), n_in tensors of the input-activation size, n_out of the output size, plus the weights"""
    e = _esz(d)
    xin = d.N * d.Hi * d.Wi * (4 if d.stem else d.Ci) * e
    yout = d.N * d.Ho * d.Wo * d.Co * e
    w = (d.Co * d.R * d.S * d.Ci * e) if wbytes is None else wbytes
    return n_in * xin + n_out * yout + w


def conv_flops(d):
    """algorithmic FLOPs of one conv pass (2 x MACs; the stem counts its real 7x7x3 taps)"""
    k = 147 if d.stem else d.R * d.S * d.Ci
    return 2.0 * d.N * d.Ho * d.Wo * d.Co * k


def conv_fwd(d, x, w, y, in_scale=None, in_shift=None, in_relu=False, bias=None, out_f32=False, stat_partial=None):
    _timed(_igemm_label(d.dtype), conv_flops(d), x, lambda: check(
        _lib.lib().frx_conv_fwd(_dev(x), _stream(x), C.byref(d), _p(x), _p(w), _p(in_scale), _p(in_shift),
                                int(in_relu), _p(bias), _p(y), int(out_f32), _p(stat_partial)), "frx_conv_fwd"),
        nbytes=conv_bytes(d))
    return y


def bn_tot(totals, replicas, count, gamma, beta=None, mean=None, invstd=None, eps=1e-5):
    """frx_bn_tot: one BatchNorm layer's statistics as replicated totals [replicas][2][C] (the tensors must outlive it)"""
    return _lib.BnTot(_pv(totals), _pv(gamma), _pv(beta), _pv(mean), _pv(invstd), int(replicas), float(count), float(eps), 0)


def conv_fwd_tot(d, x, w, y, in_bn=None, in_relu=True, stat_totals=None, stat_replicas=0):
    """conv_fwd with the prologue constants derived from `in_bn` (a BnTot) and the statistics of y added into stat_totals"""
    _timed(_igemm_label(d.dtype), conv_flops(d), x, lambda: check(
        _lib.lib().frx_conv_fwd_tot(_dev(x), _stream(x), C.byref(d), _p(x), _p(w), C.byref(in_bn) if in_bn is not None else None,
                                    int(in_relu), _p(y), _p(stat_totals), int(stat_replicas)), "frx_conv_fwd_tot"),
        nbytes=conv_bytes(d))
    return y


def conv_fwd_merge(d, y3, idn, w, y, block_out, mask=None, s3=None, b3=None, sd=None, bd=None, bn3=None, bnd=None,
                   stat_partial=None, stat_totals=None, stat_replicas=0):
    """frx_conv_fwd_merge: a 1x1 conv on relu(bn3(y3) + idn') with the merge evaluated as its prologue; block_out / mask
    receive what frx_block_merge_fwd* would have written.  BatchNorm constants as arrays (s3, b3[, sd, bd]) or totals."""
    _timed(_igemm_label(d.dtype), conv_flops(d), y3, lambda: check(
        _lib.lib().frx_conv_fwd_merge(_dev(y3), _stream(y3), C.byref(d), _p(y3), _p(idn), _p(w), _p(s3), _p(b3), _p(sd), _p(bd),
                                      C.byref(bn3) if bn3 is not None else None, C.byref(bnd) if bnd is not None else None,
                                      _p(block_out), _p(mask), _p(y), _p(stat_partial), _p(stat_totals), int(stat_replicas)),
        "frx_conv_fwd_merge"), nbytes=conv_bytes(d, n_in=3) + (0 if mask is None else mask.numel()))
    return y


def conv_fwd_keep(d, x, w, y, x_norm_out, in_scale=None, in_shift=None, in_bn=None, in_relu=True, stat_partial=None,
                  stat_totals=None, stat_replicas=0):
    """frx_conv_fwd_keep: the forward of a patch-mode 3x3 layer that also stores its prologue's output relu(bn(x))"""
    _timed(_igemm_label(d.dtype), conv_flops(d), x, lambda: check(
        _lib.lib().frx_conv_fwd_keep(_dev(x), _stream(x), C.byref(d), _p(x), _p(w), _p(in_scale), _p(in_shift),
                                     C.byref(in_bn) if in_bn is not None else None, int(in_relu), _p(y), _p(stat_partial),
                                     _p(stat_totals), int(stat_replicas), _p(x_norm_out)), "frx_conv_fwd_keep"),
        nbytes=conv_bytes(d))
    return y


def conv_dgrad(d, dy, w_crsk, dx, addend=None):
    _timed(_igemm_label(d.dtype), conv_flops(d), dy, lambda: check(
        _lib.lib().frx_conv_dgrad(_dev(dy), _stream(dy), C.byref(d), _p(dy), _p(w_crsk), _p(addend), _p(dx)),
        "frx_conv_dgrad"), nbytes=conv_bytes(d, n_in=1 + (addend is not None)))
    return dx


def conv_wgrad(d, x, dy, dw, in_scale=None, in_shift=None, in_relu=False):
    bt = 64 if (d.Co <= 64 or (32 if d.stem else d.Ci) <= 64) else 128
    _timed(f"k_wgrad<{_dt_name(d.dtype)},{bt}>", conv_flops(d), x, lambda: check(
        _lib.lib().frx_conv_wgrad(_dev(x), _stream(x), C.byref(d), _p(x), _p(in_scale), _p(in_shift),
                                  int(in_relu), _p(dy), _p(dw)), "frx_conv_wgrad"),
        nbytes=conv_bytes(d, wbytes=dw.numel() * 4))
    return dw


class WgradGroup:
    """a planned frx_wgrad_group table (device) plus everything that must outlive it"""
    __slots__ = ("table", "njobs", "nitems", "small_tiles", "dtype", "flops", "nbytes", "keep", "finish")


def wgrad_group_plan(dtype, jobs):
    """jobs: dicts with d (ConvDesc), x, dy, dw and optionally in_scale / in_shift / in_relu / pro_y / pro_coef -- or, for a
    DECOMPOSED job (include/frx.h: frx_wgrad_job.gram), gram / xsum buffers plus wk (the kernel-format weight) and coef (the
    BatchNorm-backward coefficients [3][Co]): the group then carries the table of its closing launch (wgrad_group_run runs it).
    The tensors' addresses are captured: they must stay allocated (and in place) while the group is used."""
    arr = (_lib.WgradJob * len(jobs))()
    fin, blk = [], 0
    for i, j in enumerate(jobs):
        arr[i] = _lib.WgradJob(j["d"], _pv(j["x"]), _pv(j.get("in_scale")), _pv(j.get("in_shift")),
                               int(bool(j.get("in_relu", False))), _pv(j["dy"]), _pv(j.get("pro_y")),
                               _pv(j.get("pro_coef")), _pv(j["dw"]), _pv(j.get("gram")), _pv(j.get("xsum")))
        if j.get("gram") is not None:
            d = j["d"]
            assert j["gram"].numel() == d.Ci * d.Ci + 1 and j["xsum"].numel() == d.Ci
            fin.append((j["dw"].data_ptr(), j["wk"].data_ptr(), j["gram"].data_ptr(), j["xsum"].data_ptr(), j["coef"].data_ptr(),
                        d.Co, d.Ci, blk))
            blk += (d.Co * d.Ci + 255) // 256
    nbytes = _lib.lib().frx_wgrad_group_bytes(arr, len(jobs))
    if nbytes < 0:
        raise FrxError("frx_wgrad_group_bytes: " + _lib.lib().frx_last_error().decode())
    dev = jobs[0]["x"].device
    g = WgradGroup()
    g.table = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    # the table is assembled in pinned host memory and copied by one stream-ordered asynchronous copy (enqueue-only,
    # like every libfrx call); the host image has to outlive that copy, so it stays with the group
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    n, small, nl = C.c_int(0), C.c_int(0), C.c_int(0)
    check(_lib.lib().frx_wgrad_group_plan(_dev(g.table), _stream(g.table), arr, len(jobs), C.c_void_p(host.data_ptr()),
                                          _p(g.table), nbytes, C.byref(n), C.byref(small), C.byref(nl)), "frx_wgrad_group_plan")
    g.njobs, g.nitems, g.small_tiles, g.dtype = nl.value, n.value, small.value, dtype      # (njobs: the table's layer count)
    g.flops = sum(conv_flops(j["d"]) for j in jobs)
    g.nbytes = sum(conv_bytes(j["d"], n_out=2 if j.get("pro_y") is not None else 1, wbytes=j["dw"].numel() * 4) for j in jobs)
    g.keep = [t for j in jobs for t in j.values() if isinstance(t, torch.Tensor)] + [host]
    g.finish = None
    if fin:
        g.finish = (torch.tensor(fin, dtype=torch.int64, device=dev), len(fin), blk)
    return g


def wgrad_group_run(g):
    """the list's launch and, if it holds decomposed jobs, their closing launch (frx_wgrad_gram_finish: the BatchNorm-backward
    coefficient arrays named at plan time must be final by now)"""
    def run():
        check(_lib.lib().frx_wgrad_group_run(_dev(g.table), _stream(g.table), g.dtype, _p(g.table), g.njobs, g.nitems,
                                             g.small_tiles), "frx_wgrad_group_run")
        if g.finish is not None:
            tab, n, blocks = g.finish
            check(_lib.lib().frx_wgrad_gram_finish(_dev(tab), _stream(tab), g.dtype, n, _p(tab), blocks), "frx_wgrad_gram_finish")
    _timed(f"k_wgrad_grouped<{_dt_name(g.dtype)}>", g.flops, g.table, run, nbytes=g.nbytes)


def conv_dgrad_stat_rows(d):
    r = _lib.lib().frx_conv_dgrad_stat_rows(C.byref(d))
    if r < 0:
        raise FrxError("conv descriptor rejected: " + _lib.lib().frx_last_error().decode())
    return r


def conv_dgrad_bn(d, dz, w_crsk, dx, addend=None, pro_y=None, pro_coef=None, epi_y=None, epi_out=None, epi_scale=None,
                  epi_shift=None, epi_mean=None, epi_invstd=None, epi_partial=None, pro_dy_out=None, addend_stride=0,
                  epi_out_bits=None, pro_tot=None, epi_totals=None, epi_replicas=0):
    """dgrad with the BatchNorm backward fused in (prologue: dy = alpha*dz + beta*pro_y + gam; epilogue: mask +
    per-channel reduce of the produced gradient).  pro_tot (a BnTot) instead of pro_coef / epi_totals [R][2][Ci] instead
    of epi_partial: the replicated-totals form of the statistics (csrc/bn_tot.h)."""
    f = _lib.DgradFuse(*[0 if t is None else t.data_ptr() for t in
                         (pro_y, pro_coef, epi_y, epi_out, epi_scale, epi_shift, epi_mean, epi_invstd, epi_partial, epi_out_bits)],
                       int(addend_stride), 0 if pro_dy_out is None else pro_dy_out.data_ptr(),
                       C.pointer(pro_tot) if pro_tot is not None else None,
                       0 if epi_totals is None else epi_totals.data_ptr(), int(epi_replicas))
    _timed(_igemm_label(d.dtype), conv_flops(d), dz, lambda: check(
        _lib.lib().frx_conv_dgrad_bn(_dev(dz), _stream(dz), C.byref(d), _p(dz), _p(w_crsk), _p(addend), _p(dx),
                                     C.byref(f)), "frx_conv_dgrad_bn"),
        nbytes=conv_bytes(d, n_in=1 + (addend is not None) + (epi_y is not None) + (epi_out is not None) + (epi_out_bits is not None) / 16,
                          n_out=1 + (pro_y is not None) + (pro_dy_out is not None)))
    return dx


def conv_wgrad_bn(d, x, dz, pro_y, pro_coef, dw, in_scale=None, in_shift=None, in_relu=False):
    bt = 64 if (d.Co <= 64 or d.Ci <= 64) else 128
    _timed(f"k_wgrad<{_dt_name(d.dtype)},{bt}>", conv_flops(d), x, lambda: check(
        _lib.lib().frx_conv_wgrad_bn(_dev(x), _stream(x), C.byref(d), _p(x), _p(in_scale), _p(in_shift), int(in_relu),
                                     _p(dz), _p(pro_y), _p(pro_coef), _p(dw)), "frx_conv_wgrad_bn"),
        nbytes=conv_bytes(d, n_out=2, wbytes=dw.numel() * 4))
    return dw


def bn_finalize(partial, rows, Cc, count, gamma, beta, rmean, rvar, mean, invstd, scale, shift, eps=1e-5, momentum=0.1):
    check(_lib.lib().frx_bn_finalize(_dev(partial), _stream(partial), _p(partial), rows, Cc, count, _p(gamma), _p(beta),
                                     eps, momentum, _p(rmean), _p(rvar), _p(mean), _p(invstd), _p(scale), _p(shift)),
          "frx_bn_finalize")


def bn_eval_affine(gamma, beta, rmean, rvar, scale, shift, eps=1e-5):
    check(_lib.lib().frx_bn_eval_affine(_dev(gamma), _stream(gamma), gamma.numel(), _p(gamma), _p(beta), _p(rmean),
                                        _p(rvar), eps, _p(scale), _p(shift)), "frx_bn_eval_affine")


def block_merge_fwd(dtype, rows, Cc, y3, s3, b3, idn, out, sd=None, bd=None, mask=None):
    """mask: optional uint8 [rows * Cc / V] (V = 8 bf16 / 4 fp32): the out > 0 bits, for conv_dgrad_bn(epi_out_bits=...)"""
    if mask is None:
        check(_lib.lib().frx_block_merge_fwd(_dev(y3), _stream(y3), dtype, rows, Cc, _p(y3), _p(s3), _p(b3), _p(idn),
                                             _p(sd), _p(bd), _p(out)), "frx_block_merge_fwd")
    else:
        check(_lib.lib().frx_block_merge_fwd_mask(_dev(y3), _stream(y3), dtype, rows, Cc, _p(y3), _p(s3), _p(b3), _p(idn),
                                                  _p(sd), _p(bd), _p(out), _p(mask)), "frx_block_merge_fwd_mask")
    return out


def block_merge_fwd_tot(dtype, rows, Cc, y3, bn3, idn, out, bnd=None, mask=None):
    check(_lib.lib().frx_block_merge_fwd_tot(_dev(y3), _stream(y3), dtype, rows, Cc, _p(y3), C.byref(bn3), _p(idn),
                                             C.byref(bnd) if bnd is not None else None, _p(out), _p(mask)), "frx_block_merge_fwd_tot")
    return out


def bn_finalize_batched(table, n, total_blocks):
    check(_lib.lib().frx_bn_finalize_batched(_dev(table), _stream(table), n, _p(table), total_blocks), "frx_bn_finalize_batched")


def bn_bwd_finalize_batched(table, n, total_blocks):
    check(_lib.lib().frx_bn_bwd_finalize_batched(_dev(table), _stream(table), n, _p(table), total_blocks), "frx_bn_bwd_finalize_batched")


def bn_bwd_reduce_tot(dtype, rows, Cc, g, y, mean, invstd, totals, replicas, out=None, scale=None, shift=None, relu=False,
                      dz_out=None, g_pool_hw=0):
    check(_lib.lib().frx_bn_bwd_reduce_tot(_dev(g), _stream(g), dtype, rows, Cc, _p(g), _p(y), _p(out), _p(scale), _p(shift),
                                           int(relu), _p(mean), _p(invstd), _p(dz_out), _p(totals), int(replicas), int(g_pool_hw)),
          "frx_bn_bwd_reduce_tot")


def bn_bwd_apply_tot(dtype, rows, Cc, g, y, bn, dy, out=None, scale=None, shift=None, relu=False):
    check(_lib.lib().frx_bn_bwd_apply_tot(_dev(g), _stream(g), dtype, rows, Cc, _p(g), _p(y), _p(out), _p(scale), _p(shift),
                                          int(relu), C.byref(bn), _p(dy)), "frx_bn_bwd_apply_tot")
    return dy


def stem_pool_fwd_tot(dtype, N, H, W, Cc, y, bn, out, argmax):
    check(_lib.lib().frx_stem_pool_fwd_tot(_dev(y), _stream(y), dtype, N, H, W, Cc, _p(y), C.byref(bn), _p(out), _p(argmax)),
          "frx_stem_pool_fwd_tot")


def stem_bwd_reduce_tot(dtype, N, H, W, Cc, dout, argmax, y, scale, shift, mean, invstd, totals, replicas):
    check(_lib.lib().frx_stem_bwd_reduce_tot(_dev(dout), _stream(dout), dtype, N, H, W, Cc, _p(dout), _p(argmax), _p(y), _p(scale),
                                             _p(shift), _p(mean), _p(invstd), _p(totals), int(replicas)), "frx_stem_bwd_reduce_tot")


def stem_bwd_apply_tot(dtype, N, H, W, Cc, dout, argmax, y, scale, shift, bn, dy):
    check(_lib.lib().frx_stem_bwd_apply_tot(_dev(dout), _stream(dout), dtype, N, H, W, Cc, _p(dout), _p(argmax), _p(y), _p(scale),
                                            _p(shift), C.byref(bn), _p(dy)), "frx_stem_bwd_apply_tot")
    return dy


def bn_bwd_partial_rows(rows, Cc):
    return _lib.lib().frx_bn_bwd_partial_rows(rows, Cc)


def bn_bwd_reduce(dtype, rows, Cc, g, y, mean, invstd, partial, out=None, scale=None, shift=None, relu=False, dz_out=None,
                  g_pool_hw=0):
    """g_pool_hw > 0: g is the [rows / g_pool_hw, C] gradient of an average pool, broadcast (and divided) on the fly"""
    check(_lib.lib().frx_bn_bwd_reduce(_dev(g), _stream(g), dtype, rows, Cc, _p(g), _p(y), _p(out), _p(scale),
                                       _p(shift), int(relu), _p(mean), _p(invstd), _p(dz_out), _p(partial), int(g_pool_hw)),
          "frx_bn_bwd_reduce")


def bn_bwd_finalize(partial, nblk, Cc, count, gamma, mean, invstd, dgamma, dbeta, coef):
    check(_lib.lib().frx_bn_bwd_finalize(_dev(partial), _stream(partial), _p(partial), nblk, Cc, count, _p(gamma),
                                         _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(coef)), "frx_bn_bwd_finalize")


def bn_bwd_apply(dtype, rows, Cc, g, y, mean, invstd, coef, dy, out=None, scale=None, shift=None, relu=False):
    check(_lib.lib().frx_bn_bwd_apply(_dev(g), _stream(g), dtype, rows, Cc, _p(g), _p(y), _p(out), _p(scale),
                                      _p(shift), int(relu), _p(mean), _p(invstd), _p(coef), _p(dy)), "frx_bn_bwd_apply")
    return dy


def stem_pool_fwd(dtype, N, H, W, Cc, y, scale, shift, out, argmax):
    check(_lib.lib().frx_stem_pool_fwd(_dev(y), _stream(y), dtype, N, H, W, Cc, _p(y), _p(scale), _p(shift), _p(out),
                                       _p(argmax)), "frx_stem_pool_fwd")


def stem_bwd_partial_rows():
    return _lib.lib().frx_stem_bwd_partial_rows()


def stem_bwd_reduce(dtype, N, H, W, Cc, dout, argmax, y, scale, shift, mean, invstd, partial):
    """pool gather -> ReLU mask -> (sum dz, sum dz*xhat) partials, without the full-resolution gradient in memory"""
    check(_lib.lib().frx_stem_bwd_reduce(_dev(dout), _stream(dout), dtype, N, H, W, Cc, _p(dout), _p(argmax), _p(y), _p(scale),
                                         _p(shift), _p(mean), _p(invstd), _p(partial)), "frx_stem_bwd_reduce")


def stem_bwd_apply(dtype, N, H, W, Cc, dout, argmax, y, scale, shift, coef, dy):
    check(_lib.lib().frx_stem_bwd_apply(_dev(dout), _stream(dout), dtype, N, H, W, Cc, _p(dout), _p(argmax), _p(y), _p(scale),
                                        _p(shift), _p(coef), _p(dy)), "frx_stem_bwd_apply")
    return dy


def stem_pool_bwd(dtype, N, H, W, Cc, dout, argmax, dpost):
    check(_lib.lib().frx_stem_pool_bwd(_dev(dout), _stream(dout), dtype, N, H, W, Cc, _p(dout), _p(argmax), _p(dpost)),
          "frx_stem_pool_bwd")


def avgpool_fwd(dtype, N, HW, Cc, x, out):
    check(_lib.lib().frx_avgpool_fwd(_dev(x), _stream(x), dtype, N, HW, Cc, _p(x), _p(out)), "frx_avgpool_fwd")


def avgpool_bwd(dtype, N, HW, Cc, dpool, dx):
    check(_lib.lib().frx_avgpool_bwd(_dev(dpool), _stream(dpool), dtype, N, HW, Cc, _p(dpool), _p(dx)), "frx_avgpool_bwd")


def sgd_step(p, g, buf, lr, momentum=0.9, weight_decay=5e-4, grad_scale=1.0, lr_dev=None):
    check(_lib.lib().frx_sgd_step(_dev(p), _stream(p), p.numel(), _p(p), _p(g), _p(buf), _p(lr_dev), float(lr),
                                  float(momentum), float(weight_decay), float(grad_scale)), "frx_sgd_step")


def sgd_step_prep(dtype, table, total_blocks, p, g, buf, lr, momentum=0.9, weight_decay=5e-4, grad_scale=1.0, lr_dev=None,
                  zero_grads=True):
    """SGD + kernel-format weight copies (+ the gradient zero-fill) in one launch (include/frx.h: frx_sgd_step_prep)"""
    check(_lib.lib().frx_sgd_step_prep(_dev(p), _stream(p), dtype, table.shape[0], _p(table), total_blocks, _p(p), _p(g), _p(buf),
                                       _p(lr_dev), float(lr), float(momentum), float(weight_decay), float(grad_scale),
                                       1 if zero_grads else 0), "frx_sgd_step_prep")


def weight_prep(dtype, Co, RS, Ci, master, krsc=None, crsk=None):
    check(_lib.lib().frx_weight_prep(_dev(master), _stream(master), dtype, Co, RS, Ci, _p(master), _p(krsc), _p(crsk)),
          "frx_weight_prep")


def weight_prep_batched(dtype, table, master, total_blocks):
    check(_lib.lib().frx_weight_prep_batched(_dev(master), _stream(master), dtype, table.shape[0], _p(table), _p(master),
                                             total_blocks), "frx_weight_prep_batched")


def input_prep(dtype, images, out):
    """images: fp32 [N,3,H,W] in [-1,1] or uint8 [N,H,W,3], contiguous; out: the stem's zero-bordered NHWC4 buffer
    planned for exactly this N, H, W (the C ABI rejects any other size instead of writing past it)"""
    if images.dim() != 4:
        raise FrxError(f"input_prep: expected a 4-d image batch, got shape {tuple(images.shape)}")
    if images.dtype == torch.uint8:
        _chk(images, torch.uint8, "images")
        N, H, W, ch = images.shape
        u8 = 1
    else:
        _chk(images, torch.float32, "images")
        N, ch, H, W = images.shape
        u8 = 0
    if ch != 3:
        raise FrxError(f"input_prep: expected 3 colour channels ({'NHWC uint8' if u8 else 'NCHW fp32'}), got shape {tuple(images.shape)}")
    _chk(out, TORCH_DT[dtype], "input_prep destination")
    check(_lib.lib().frx_input_prep(_dev(images), _stream(images), dtype, N, H, W, _p(images), u8, _p(out), out.numel()),
          "frx_input_prep")
    return out


def cast(dtype, x, y, to_f32):
    check(_lib.lib().frx_cast(_dev(x), _stream(x), dtype, int(to_f32), x.numel(), _p(x), _p(y)), "frx_cast")
    return y


def colsum_f32(x, out):
    check(_lib.lib().frx_colsum_f32(_dev(x), _stream(x), x.shape[0], x.shape[1], _p(x), _p(out)), "frx_colsum_f32")
