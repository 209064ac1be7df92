"""Builds the gfx950 HIP libraries with hipcc, in-tree:

    qqq_amd/libqqq_amd.so      the operator (include/qqq_amd.h)      <- csrc/qqq_w4a8.hip (+ csrc/*.hip.h)
    qqq_amd/libqqq_amd_dev.so  test / tuning companion (include/qqq_amd_dev.h) <- csrc/qqq_dev.hip
    qqq_amd/_torch_ext*.so     compiled torch binding (pybind + TORCH_LIBRARY) of the operator library <- csrc/qqq_torch.cpp

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


TORCH_SRC = os.path.join(_HERE, "csrc", "qqq_torch.cpp")


def torch_ext_path() -> str:
    import sysconfig

    return os.path.join(_HERE, "_torch_ext" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    """g++ against torch's own headers (a ROCm torch ships c10/hip/*.h): no kernel code, no hipify -- the module only forwards
    data_ptr()s and the current HIP stream to libqqq_amd.so, which it finds next to itself ($ORIGIN)."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce

    out = torch_ext_path()
    default_lib = os.path.join(_HERE, "libqqq_amd.so")
    if not os.path.exists(default_lib):
        raise RuntimeError("build the operator library first (qqq_amd.build.build())")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(TORCH_SRC), os.path.getmtime(HDR)):
        return out
    tdir = os.path.dirname(torch.__file__)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_EXTENSION_NAME=_torch_ext",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include"]
    cmd += [TORCH_SRC, "-o", out + ".tmp", f"-L{tdir}/lib", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_python",
            f"-L{_HERE}", "-lqqq_amd", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tdir}/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_dev(force=True, verbose=True))
    print(build_torch_ext(force=True, verbose=True))
