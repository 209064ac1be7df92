"""Builds qqq_amd/libqqq_amd.so (hand-written HIP for gfx950) with hipcc, in-tree.

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "qqq_w4a8.hip")  # the one translation unit; it includes csrc/*.hip.h
HDR = os.path.join(os.path.dirname(_HERE), "include", "qqq_amd.h")
LIB = os.environ.get("QQQ_AMD_LIB") or os.path.join(_HERE, "libqqq_amd.so")  # override: tuning builds only
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    csrc = os.path.dirname(SRC)
    deps = [HDR] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [
        hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-Wall", "-Wno-unused-function", "-o", LIB + ".tmp", SRC,
    ] + os.environ.get("QQQ_AMD_CXXFLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
