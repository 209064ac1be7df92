"""ctypes binding of the test / tuning companion library (include/qqq_amd_dev.h, libqqq_amd_dev.so).

Used by tests/, bench.py and tools/ only -- the operator (qqq_amd.ops, qqq_amd.qlinear) never imports this module."""
from __future__ import annotations

import ctypes
import os

from . import _lib
from . import build as _build

_dev = None


def lib():
    global _dev
    if _dev is not None:
        return _dev
    _lib.lib()  # the operator library (and the HIP runtime torch brought in) first
    path = _build.DEV_LIB
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python qqq_amd/build.py` (needs hipcc)")
    L = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.qqq_dev_probe_mfma.argtypes = [ci, vp, vp, vp, ci, vp]
    L.qqq_dev_probe_mfma.restype = ci
    L.qqq_dev_probe_glds.argtypes = [vp, vp, vp, ci, vp]
    L.qqq_dev_probe_glds.restype = ci
    L.qqq_dev_probe_dequant.argtypes = [vp, vp, vp, vp, ci, ci, vp]
    L.qqq_dev_probe_dequant.restype = ci
    L.qqq_dev_probe_fill.argtypes = [vp, ctypes.c_size_t, ctypes.c_size_t, ci, ci, ci, vp, ci, vp, ctypes.POINTER(ctypes.c_float)]
    L.qqq_dev_probe_fill.restype = ci
    L.qqq_dev_probe_mfma_rate.argtypes = [ci, vp, ci, ci, vp, ci, vp, ctypes.POINTER(ctypes.c_float)]
    L.qqq_dev_probe_mfma_rate.restype = ci
    L.qqq_dev_bench_gemm.argtypes = [vp, vp, ctypes.POINTER(vp), ci, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, ci,
                                     ctypes.POINTER(_lib.QQQTune), ci, ctypes.POINTER(ctypes.c_float)]
    L.qqq_dev_bench_gemm.restype = ci
    L.qqq_dev_bench_gemm2.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, ci,
                                      ctypes.POINTER(_lib.QQQTune), ci, ctypes.POINTER(ctypes.c_float)]
    L.qqq_dev_bench_gemm2.restype = ci
    L.qqq_dev_probe_placement.argtypes = [ctypes.POINTER(ctypes.c_uint32), ci, ci, ci, vp, ci, vp]
    L.qqq_dev_probe_placement.restype = ci
    L.qqq_dev_last_error.restype = ctypes.c_char_p
    _dev = L
    return L


def gemm_ex_ptr(operator_lib=None):
    """address of qqq_w4a8_gemm_ex in the operator library (what qqq_dev_bench_gemm times)"""
    L = operator_lib if operator_lib is not None else _lib.lib()
    return ctypes.cast(L.qqq_w4a8_gemm_ex, ctypes.c_void_p)


def gemm_ex2_ptr(operator_lib=None):
    """address of qqq_w4a8_gemm_ex2 (what qqq_dev_bench_gemm2 times)"""
    L = operator_lib if operator_lib is not None else _lib.lib()
    return ctypes.cast(L.qqq_w4a8_gemm_ex2, ctypes.c_void_p)


def last_error() -> str:
    return lib().qqq_dev_last_error().decode()
