"""Builds the gfx950 HIP libraries with hipcc, in-tree:

    qqq_amd/libqqq_amd.so      the operator (include/qqq_amd.h)      <- csrc/qqq_w4a8.hip (+ csrc/*.hip.h)
    qqq_amd/libqqq_amd_dev.so  test / tuning companion (include/qqq_amd_dev.h) <- csrc/qqq_dev.hip

The .so files are git-ignored but travel to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "qqq_w4a8.hip")  # the one translation unit of the operator; it includes csrc/*.hip.h
DEV_SRC = os.path.join(_HERE, "csrc", "qqq_dev.hip")
HDR = os.path.join(os.path.dirname(_HERE), "include", "qqq_amd.h")
DEV_HDR = os.path.join(os.path.dirname(_HERE), "include", "qqq_amd_dev.h")
LIB = os.environ.get("QQQ_AMD_LIB") or os.path.join(_HERE, "libqqq_amd.so")  # override: tuning builds only
DEV_LIB = os.path.join(_HERE, "libqqq_amd_dev.so")
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def _stale(lib: str, extra) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    csrc = os.path.dirname(SRC)
    deps = list(extra) + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    return any(os.path.getmtime(p) > t for p in deps)


def needs_build() -> bool:
    return _stale(LIB, [HDR])


def _compile(src: str, out: str, verbose: bool, flags=()) -> str:
    cmd = [
        hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
        "-Wall", "-Wno-unused-function", "-o", out + ".tmp", src,
    ] + list(flags) + os.environ.get("QQQ_AMD_CXXFLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    return _compile(SRC, LIB, verbose)


def build_dev(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale(DEV_LIB, [HDR, DEV_HDR]):
        return DEV_LIB
    return _compile(DEV_SRC, DEV_LIB, verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_dev(force=True, verbose=True))
