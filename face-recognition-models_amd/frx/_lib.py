"""ctypes loader for libfrx.so (include/frx.h).  The library is built in-tree by
`make -C face-recognition-models_amd/csrc` (or __graft_entry__.build())."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("FRX_LIB") or os.path.join(_HERE, "libfrx.so")      # (FRX_LIB: another build of the same library -- scripts/p3_ablate.sh)
_lib = None


class FrxError(RuntimeError):
    pass


class HeadDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("N", C.c_int32), ("D", C.c_int32), ("C", C.c_int32),
                ("s", C.c_float), ("m", C.c_float), ("momentum", C.c_float), ("lamb", C.c_float),
                ("p", C.c_float * 4), ("flags", C.c_int32), ("class_offset", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "N", "Hi", "Wi", "Ci", "Co", "R", "S", "stride", "pad",
                                          "Ho", "Wo", "stem")]


class BnTot(C.Structure):
    """frx_bn_tot (include/frx.h): BatchNorm statistics as replicated totals"""
    _fields_ = ([(n, C.c_void_p) for n in ("totals", "gamma", "beta", "mean", "invstd")]
                + [("replicas", C.c_int32), ("count", C.c_float), ("eps", C.c_float), ("reserved", C.c_int32)])


class DgradFuse(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("pro_y", "pro_coef", "epi_y", "epi_out", "epi_scale", "epi_shift",
                                           "epi_mean", "epi_invstd", "epi_partial", "epi_out_bits")]
                + [("addend_stride", C.c_int32), ("pro_dy_out", C.c_void_p), ("pro_tot", C.POINTER(BnTot)),
                   ("epi_totals", C.c_void_p), ("epi_replicas", C.c_int32)])


class WgradJob(C.Structure):
    """frx_wgrad_job (include/frx.h)"""
    _fields_ = [("d", ConvDesc), ("x", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
                ("in_relu", C.c_int32), ("dy", C.c_void_p), ("pro_y", C.c_void_p), ("pro_coef", C.c_void_p),
                ("dw", C.c_void_p), ("gram", C.c_void_p), ("xsum", C.c_void_p)]


def library_path() -> str:
    return _LIB_PATH


_P = C.c_void_p
_SIGS = {
    "frx_version": (C.c_int, []),
    "frx_struct_sizes": (C.c_int, [C.POINTER(C.c_int64)]),
    "frx_last_error": (C.c_char_p, []),
    "frx_device_props": (C.c_int, [C.c_int, C.POINTER(C.c_int64)]),
    "frx_head_workspace_bytes": (C.c_size_t, [C.POINTER(HeadDesc)]),
    "frx_head_fwd_cos": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, C.c_size_t, _P]),
    "frx_head_fwd_loss": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, C.c_int64, _P, C.c_size_t,
                                    _P, _P, _P, _P, _P, _P]),
    "frx_head_fwd": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, _P, C.c_size_t,
                               _P, _P, _P, _P, _P, _P]),
    "frx_head_bwd": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, _P, _P, C.c_size_t,
                               _P, _P, C.c_int]),
    "frx_head_target_cos": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, C.c_size_t, _P]),
    "frx_head_shard_cos": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, C.c_size_t, _P]),
    "frx_head_shard_rows": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, C.c_size_t, _P]),
    "frx_head_shard_rescale": (C.c_int, [C.c_int, _P, C.c_int, _P, _P, _P]),
    "frx_head_shard_finish": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    "frx_head_aux": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, C.c_size_t, _P, _P]),
    "frx_head_vpl_prepare": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, C.c_size_t]),
    "frx_conv_stat_rows": (C.c_int, [C.POINTER(ConvDesc)]),
    "frx_stem_padded_dims": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "frx_conv_fwd": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, _P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "frx_conv_fwd_tot": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, C.POINTER(BnTot), C.c_int, _P, _P, C.c_int]),
    "frx_conv_fwd_merge": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, C.POINTER(BnTot), C.POINTER(BnTot),
                                     _P, _P, _P, _P, _P, C.c_int]),
    "frx_conv_fwd_keep": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, _P, _P, C.POINTER(BnTot), C.c_int, _P, _P, _P, C.c_int, _P]),
    "frx_block_merge_fwd_tot": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, C.POINTER(BnTot), _P, C.POINTER(BnTot), _P, _P]),
    "frx_bn_finalize_batched": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int]),
    "frx_bn_bwd_finalize_batched": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int]),
    "frx_bn_bwd_reduce_tot": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_int,
                                        _P, _P, _P, _P, C.c_int, C.c_int]),
    "frx_bn_bwd_apply_tot": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_int,
                                       C.POINTER(BnTot), _P]),
    "frx_stem_pool_fwd_tot": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(BnTot), _P, _P]),
    "frx_stem_bwd_reduce_tot": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int]),
    "frx_stem_bwd_apply_tot": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.POINTER(BnTot), _P]),
    "frx_conv_dgrad": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "frx_conv_wgrad": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, _P, C.c_int, _P, _P]),
    "frx_conv_dgrad_stat_rows": (C.c_int, [C.POINTER(ConvDesc)]),
    "frx_conv_tile": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "frx_last_conv_launch": (C.c_int, [C.POINTER(C.c_int)]),
    "frx_conv_patch_mode": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "frx_wgrad_group_bytes": (C.c_int64, [C.POINTER(WgradJob), C.c_int]),
    "frx_wgrad_group_plan": (C.c_int, [C.c_int, _P, C.POINTER(WgradJob), C.c_int, _P, _P, C.c_int64, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "frx_wgrad_gram_finish": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, C.c_int]),
    "frx_wgrad_group_run": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "frx_conv_dgrad_bn": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, _P, _P, C.POINTER(DgradFuse)]),
    "frx_conv_wgrad_bn": (C.c_int, [C.c_int, _P, C.POINTER(ConvDesc), _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "frx_bn_finalize": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int64, _P, _P, C.c_float, C.c_float,
                                  _P, _P, _P, _P, _P, _P]),
    "frx_bn_eval_affine": (C.c_int, [C.c_int, _P, C.c_int, _P, _P, _P, _P, C.c_float, _P, _P]),
    "frx_block_merge_fwd": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "frx_block_merge_fwd_mask": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frx_bn_bwd_partial_rows": (C.c_int, [C.c_int64, C.c_int]),
    "frx_bn_bwd_reduce": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_int,
                                    _P, _P, _P, _P, C.c_int]),
    "frx_bn_bwd_finalize": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "frx_bn_bwd_apply": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_int,
                                   _P, _P, _P, _P]),
    "frx_stem_pool_fwd": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "frx_stem_bwd_partial_rows": (C.c_int, []),
    "frx_stem_bwd_reduce": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frx_stem_bwd_apply": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "frx_stem_pool_bwd": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "frx_avgpool_fwd": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "frx_avgpool_bwd": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "frx_sgd_step": (C.c_int, [C.c_int, _P, C.c_int64, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float]),
    "frx_weight_prep": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "frx_weight_prep_batched": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int]),
    "frx_sgd_step_prep": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_int]),
    "frx_input_prep": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int64]),
    "frx_cast": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int64, _P, _P]),
    "frx_colsum_f32": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "frx_head_bwd_dlogits": (C.c_int, [C.c_int, _P, C.POINTER(HeadDesc), _P, _P, _P, _P, _P, _P, C.c_size_t,
                                       _P, _P, C.c_int]),
    "frx_pair_cosine": (C.c_int, [C.c_int, _P, _P, _P, C.c_int64, C.c_int32, _P]),
    "frx_threshold_count": (C.c_int, [C.c_int, _P, _P, _P, C.c_int64, C.c_float, _P]),
}


def load_library(path: str | None = None):
    """Load libfrx.so and bind every symbol include/frx.h declares.  Raises FrxError
    (never falls back) when the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch bundles its own HIP runtime; it must be in the process first so libfrx.so binds to the
    # SAME runtime (two runtimes in one process do not share devices, streams or allocations)
    import torch  # noqa: F401
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise FrxError(f"native library not built: {p} is missing -- run "
                       f"`make -C face-recognition-models_amd/csrc` or __graft_entry__.build()")
    try:
        lib_ = C.CDLL(p)
    except OSError as e:  # e.g. ROCm runtime absent
        raise FrxError(f"cannot load {p}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib_, name)
        except AttributeError as e:
            raise FrxError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib_
    return _lib


def lib():
    return load_library()


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().frx_last_error().decode("utf-8", "replace")
        raise FrxError(f"{what} failed (status {rc}): {msg}")


def exported_symbols():
    return list(_SIGS)
