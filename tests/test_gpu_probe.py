"""Device checks of the hardware facts the kernels rely on: MFMA operand/accumulator lane maps and
global_load_lds (LDS-DMA) destination semantics.  Run first: if these fail, nothing else can pass."""
import ctypes

import numpy as np
import pytest
import torch

import lane_model as LM
from conftest import gpu_dump

pytestmark = pytest.mark.gpu


def _run_mfma(kind, a, b, dev):
    from qqq_amd import _dev as _lib

    L = _lib.lib()
    nreg = 4 if kind == 16 else 16
    ta = torch.from_numpy(a).to(dev)
    tb = torch.from_numpy(b).to(dev)
    out = torch.zeros((64, nreg), dtype=torch.int32, device=dev)
    rc = L.qqq_dev_probe_mfma(kind, ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()),
                          ctypes.c_void_p(out.data_ptr()), 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("kind", [16, 32])
def test_mfma_lane_maps(kind, dev):
    rng = np.random.default_rng(kind)
    a = rng.integers(-128, 128, size=(64, 16), dtype=np.int8)
    b = rng.integers(-128, 128, size=(64, 16), dtype=np.int8)  # asymmetric on purpose
    got = _run_mfma(kind, a, b, dev)
    exp = (LM.mfma_16x16x64 if kind == 16 else LM.mfma_32x32x32)(a, b)
    if not np.array_equal(got, exp):
        gpu_dump(f"probe_mfma{kind}", a=a, b=b, got=got, exp=exp)
    assert np.array_equal(got, exp)


def test_global_load_lds_is_lane_linear(dev):
    from qqq_amd import _dev as _lib

    L = _lib.lib()
    rng = np.random.default_rng(7)
    src = rng.integers(0, 2**31, size=(64, 4), dtype=np.int64).astype(np.uint32)
    perm = rng.permutation(64).astype(np.int32)
    ts = torch.from_numpy(src.view(np.int32)).to(dev)
    tp = torch.from_numpy(perm).to(dev)
    out = torch.zeros((64, 4), dtype=torch.int32, device=dev)
    rc = L.qqq_dev_probe_glds(ctypes.c_void_p(ts.data_ptr()), ctypes.c_void_p(tp.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                          0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    if not np.array_equal(got, src[perm]):
        gpu_dump("probe_glds", src=src, perm=perm, got=got)
    assert np.array_equal(got, src[perm])
