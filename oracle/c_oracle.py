"""ctypes loader for oracle/libqqq_oracle.so (plain-C CPU oracle) -- TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqqq_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "qqq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libqqq_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.qqq_oracle_unpack.argtypes = [vp, ci, ci, ci, vp]
        L.qqq_oracle_pack.argtypes = [vp, ci, ci, ci, vp]
        L.qqq_oracle_weight_operand.argtypes = [vp, vp, ci, ci, ci, vp]
        L.qqq_oracle_gemm.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]
        L.qqq_oracle_gemm.restype = ci
        L.qqq_oracle_dynamic_quant.argtypes = [vp, ci, ci, ci, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def unpack(B: np.ndarray, grouped: bool) -> np.ndarray:
    B = np.ascontiguousarray(B, dtype=np.int32)
    K, N = B.shape[0] * 16, B.shape[1] // 2
    codes = np.empty((K, N), np.int8)
    lib().qqq_oracle_unpack(_p(B), K, N, int(grouped), _p(codes))
    return codes


def pack(codes: np.ndarray, grouped: bool) -> np.ndarray:
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    K, N = codes.shape
    B = np.empty((K // 16, 2 * N), np.int32)
    lib().qqq_oracle_pack(_p(codes), K, N, int(grouped), _p(B))
    return B


def weight_operand(B, s3, groupsize: int) -> np.ndarray:
    B = np.ascontiguousarray(B, dtype=np.int32)
    K, N = B.shape[0] * 16, B.shape[1] // 2
    s3c = None if groupsize <= 0 else np.ascontiguousarray(s3).view(np.uint16)
    Wq = np.empty((K, N), np.int8)
    lib().qqq_oracle_weight_operand(_p(B), _p(s3c), K, N, groupsize, _p(Wq))
    return Wq


def qqq_gemm(A, B, s1, s2, s3=None, return_acc=False):
    A = np.ascontiguousarray(A, dtype=np.int8)
    B = np.ascontiguousarray(B, dtype=np.int32)
    M, K = A.shape
    N = B.shape[1] // 2
    assert B.shape[0] * 16 == K
    s1 = np.ascontiguousarray(s1, dtype=np.float32).reshape(-1)
    s2 = np.ascontiguousarray(s2, dtype=np.float32).reshape(-1)
    grouped = s3 is not None and np.asarray(s3).size != 0
    groupsize = -1
    s3c = None
    if grouped:
        s3c = np.ascontiguousarray(s3).view(np.uint16)
        groupsize = K // s3c.shape[0]
    acc = np.empty((M, N), np.int32) if return_acc else None
    D = np.empty((M, N), np.uint16)
    rc = lib().qqq_oracle_gemm(_p(A), _p(B), _p(s1), _p(s2), _p(s3c), M, N, K, groupsize, _p(acc), _p(D))
    if rc:
        raise RuntimeError(f"qqq_oracle_gemm rc={rc}")
    D = D.view(np.float16)
    return (D, acc) if return_acc else D


def dynamic_quant(x, scalar_div: str = "div"):
    x = np.ascontiguousarray(x, dtype=np.float16)
    M, K = x.shape
    xq = np.empty((M, K), np.int8)
    s1 = np.empty((M, 1), np.float32)
    lib().qqq_oracle_dynamic_quant(_p(x.view(np.uint16)), M, K, int(scalar_div == "recip"), _p(xq), _p(s1))
    return xq, s1
