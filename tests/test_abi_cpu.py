"""The C-ABI library loads without a GPU and exports every symbol include/qqq_amd.h declares; the
reference's shape validation (return codes 0/1/2) is reproduced before any HIP call."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from qqq_amd import _lib, build

    build.build()
    return _lib.lib()


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "qqq_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(qqq_[a-z0-9_]+)\s*\(", hdr))
    assert {"qqq_w4a8_gemm", "qqq_w4a8_gemm_ex", "qqq_dynamic_quant", "qqq_add_bias", "qqq_amd_abi_version",
            "qqq_amd_last_error", "qqq_probe_mfma", "qqq_probe_glds", "qqq_bench_gemm"} <= names
    for n in names:
        assert hasattr(L, n), n
    assert L.qqq_amd_abi_version() == 1


def _call(L, m, n, k, groupsize=-1, thread_k=-1, thread_n=-1):
    z = None  # shape validation happens before any pointer is touched
    return L.qqq_w4a8_gemm(z, z, z, z, z, z, z, m, n, k, z, groupsize, 0, z, thread_k, thread_n, -1, 16)


def test_reference_shape_validation_codes(L):
    # reference: is_valid_config / determine_thread_config / CALL_IF (csrc/qqq_gemm.cu:867-945, :990)
    assert _call(L, 16, 100, 256) == 1            # n not a multiple of 64
    assert _call(L, 16, 256, 96) == 1             # k not a multiple of 64
    assert _call(L, 16, 64, 64) == 1              # n=64 needs thread_k=128 -> k % 128
    assert _call(L, 0, 256, 256) == 0             # empty problem: success, nothing launched (:1002)
    assert _call(L, 16, 256, 0) == 0 or _call(L, 16, 256, 0) == 1
    assert _call(L, 16, 256, 256, thread_k=32, thread_n=128) == 1   # thread_k must be 64 or 128
    assert _call(L, 16, 256, 256, thread_k=64, thread_n=128) == 2   # valid tiling, no such kernel at 256 threads
    assert _call(L, 16, 256, 256, groupsize=64) == 2                # only group 128 kernels exist
    assert _call(L, 0, 256, 256, groupsize=64) == 0
    # valid shape, null pointers: our own argument check, still no launch
    assert _call(L, 16, 256, 256) == 17


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from qqq_amd import _lib, build

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
