"""Thin tensor-level wrappers over the C ABI.  torch is used for device memory and
streams only.  Every function requires HIP tensors and a loaded libfrx.so."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import FrxError, HeadDesc, check

ARC, COS, SPHERE, CURR = 0, 1, 2, 3


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise FrxError("frx ops need tensors on the HIP device (there is no CPU path)")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(t, dtype, name):
    if t.dtype != dtype or not t.is_contiguous():
        raise FrxError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")


class HeadContext:
    """Descriptor + workspace of one head instance (one per model; reused every step)."""

    def __init__(self, kind, N, D, C_, s, m, momentum=0.01, device=None):
        self.desc = HeadDesc(kind, N, D, C_, s, m, momentum, 0.0)
        nbytes = _lib.lib().frx_head_workspace_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise FrxError("head descriptor rejected: " + _lib.lib().frx_last_error().decode())
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.nbytes = nbytes

    @property
    def shape(self):
        return self.desc.N, self.desc.D, self.desc.C


def head_forward(ctx: HeadContext, x, w, labels, state_t=None, lamb=0.0, want_logits=False,
                 ty_allreduce=None):
    """Returns dict(loss[1], topk[2] int32, norms[N], lse[N], cos_s, logits).
    ty_allreduce: optional callable(tensor[1]) -> count, summing the target-cosine sum over
    data-parallel ranks between the two phases (CurricularFace EMA)."""
    N, D, Cc = ctx.shape
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "w"); _chk(labels, torch.int64, "labels")
    if tuple(x.shape) != (N, D) or labels.numel() != N or w.numel() != D * Cc:
        raise FrxError(f"head_forward: shapes x{tuple(x.shape)} w{tuple(w.shape)} labels{tuple(labels.shape)} "
                       f"do not match the context (N={N}, D={D}, C={Cc})")
    ctx.desc.lamb = float(lamb)
    dev, st = _dev(x), _stream(x)
    o = dict(loss=torch.empty(1, device=x.device), topk=torch.empty(2, dtype=torch.int32, device=x.device),
             norms=torch.empty(N, device=x.device), lse=torch.empty(N, device=x.device),
             cos_s=None, logits=None)
    if want_logits:
        o["cos_s"] = torch.empty(N, Cc, device=x.device)
        o["logits"] = torch.empty(N, Cc, device=x.device)
    L = _lib.lib()
    if ty_allreduce is None:
        check(L.frx_head_fwd(dev, st, C.byref(ctx.desc), _p(x), _p(w), _p(labels), _p(state_t), _p(ctx.ws),
                             ctx.nbytes, _p(o["cos_s"]), _p(o["logits"]), _p(o["norms"]), _p(o["loss"]),
                             _p(o["lse"]), _p(o["topk"])), "frx_head_fwd")
    else:
        tys = torch.empty(1, device=x.device)
        check(L.frx_head_fwd_cos(dev, st, C.byref(ctx.desc), _p(x), _p(w), _p(labels), _p(ctx.ws), ctx.nbytes,
                                 _p(tys)), "frx_head_fwd_cos")
        count = ty_allreduce(tys)
        check(L.frx_head_fwd_loss(dev, st, C.byref(ctx.desc), _p(labels), _p(state_t), _p(tys), int(count),
                                  _p(ctx.ws), ctx.nbytes, _p(o["cos_s"]), _p(o["logits"]), _p(o["norms"]),
                                  _p(o["loss"]), _p(o["lse"]), _p(o["topk"])), "frx_head_fwd_loss")
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
