"""ctypes binding of the C-ABI in include/qqq_amd.h.  No fallback: if the HIP library is missing or
does not load, every entry point raises -- the product path never routes through a CPU implementation."""
from __future__ import annotations

import ctypes
import os

from . import build as _build

_lib = None
ABI_VERSION = 4  # QQQ_AMD_ABI_VERSION in include/qqq_amd.h


class QQQTune(ctypes.Structure):
    _fields_ = [
        ("kernel", ctypes.c_int), ("ksplit", ctypes.c_int), ("waves", ctypes.c_int),
        ("fused", ctypes.c_int), ("bm", ctypes.c_int), ("glds", ctypes.c_int),
        ("pf", ctypes.c_int), ("stages", ctypes.c_int), ("mt", ctypes.c_int), ("pw", ctypes.c_int),
        ("nslots", ctypes.c_int), ("split_m", ctypes.c_int), ("skew", ctypes.c_int), ("w8", ctypes.c_int),
    ]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings the HIP runtime (SONAME libamdhip64.so.7) into the process, and our
    # library's NEEDED entry then binds to that same runtime (one HIP runtime per process).
    import torch  # noqa: F401

    path = _build.LIB
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the gfx950 HIP kernels are not built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback."
        )
    L = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    gemm_args = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, ci, ci, ci, ci]
    L.qqq_w4a8_gemm.argtypes = gemm_args
    L.qqq_w4a8_gemm.restype = ci
    L.qqq_w4a8_gemm_ex.argtypes = gemm_args + [ctypes.POINTER(QQQTune), vp, vp]
    L.qqq_w4a8_gemm_ex.restype = ci
    L.qqq_w4a8_gemm_ex2.argtypes = gemm_args + [ctypes.POINTER(QQQTune), vp, vp, vp]
    L.qqq_w4a8_gemm_ex2.restype = ci
    L.qqq_expand_int8.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
    L.qqq_expand_int8.restype = ci
    L.qqq_quantlinear_forward2.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, ci, vp, vp]
    L.qqq_quantlinear_forward2.restype = ci
    L.qqq_w4a8_plan.argtypes = [ci, ci, ci, ci, ci, ci, ci, ctypes.POINTER(QQQTune), ctypes.POINTER(QQQTune)]
    L.qqq_w4a8_plan.restype = ci
    L.qqq_w4a8_model_us.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    L.qqq_w4a8_model_us.restype = ci
    L.qqq_dynamic_quant.argtypes = [vp, vp, vp, ci, ci, ci, vp]
    L.qqq_dynamic_quant.restype = ci
    L.qqq_quantlinear_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, ci, vp]
    L.qqq_quantlinear_forward.restype = ci
    L.qqq_pack_int4.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    L.qqq_pack_int4.restype = ci
    L.qqq_unpack_int4.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    L.qqq_unpack_int4.restype = ci
    L.qqq_amd_abi_version.restype = ci
    L.qqq_amd_last_error.restype = ctypes.c_char_p
    if L.qqq_amd_abi_version() != ABI_VERSION:
        raise RuntimeError("libqqq_amd.so ABI version mismatch; rebuild")
    _lib = L
    return L


def last_error() -> str:
    return lib().qqq_amd_last_error().decode()


def plan(m, n, k, groupsize=-1, max_par=16, have_scratch=True, have_workspace=True, tune=None) -> dict:
    """the dispatch decision for one problem (host logic only, no GPU work): see qqq_w4a8_plan"""
    tn = None
    if tune:
        tn = QQQTune()
        for key, v in tune.items():
            setattr(tn, key, int(v))
    out = QQQTune()
    rc = lib().qqq_w4a8_plan(m, n, k, groupsize, max_par, int(have_scratch), int(have_workspace),
                             ctypes.byref(tn) if tn is not None else None, ctypes.byref(out))
    if rc:
        raise RuntimeError(f"qqq_w4a8_plan failed ({rc}): {last_error()}")
    return {f: getattr(out, f) for f, _ in QQQTune._fields_ if f != "reserved"}


def model_us(m, n, k, groupsize=-1, max_par=16) -> dict:
    """the cost models' price (us) of each family for one problem: {"column", "stream", "panel", "wide"} (families that are no candidate are left out)"""
    out = (ctypes.c_double * 4)()
    rc = lib().qqq_w4a8_model_us(m, n, k, groupsize, max_par, out)
    if rc:
        raise RuntimeError(f"qqq_w4a8_model_us failed ({rc}): {last_error()}")
    return {name: out[i] for i, name in enumerate(("column", "stream", "panel", "wide")) if out[i] > 0}
