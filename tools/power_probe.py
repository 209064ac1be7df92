#!/usr/bin/env python3
"""How much of the tiled kernel's time is operand-toggle power: same launch (M=4096, N=8192, K=21760, per-channel),
different operand statistics.  The cycle count is data-independent; only the clock the chip settles at changes."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import _dev
dev = torch.device("cuda:0")
L = _dev.lib()
M = 4096
layer = Bn.Layer(dev, grouped=False, nbuf=4)
arr = (ctypes.c_void_p * len(layer.Bs))(*[b.data_ptr() for b in layer.Bs])
A0, s1 = Bn.make_tokens(dev, M, M)
D = torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
cases = {
    "gaussian int8 (bench)": A0,
    "uniform int8 [-128,127]": torch.randint(-128, 128, A0.shape, generator=g, dtype=torch.int8, device=dev),
    "small [-8,7]": torch.randint(-8, 8, A0.shape, generator=g, dtype=torch.int8, device=dev),
    "zeros": torch.zeros_like(A0),
}
def run(A, n=6):
    out = (ctypes.c_float * n)()
    st = torch.cuda.current_stream(dev).cuda_stream
    rc = L.qqq_dev_bench_gemm(_dev.gemm_ex_ptr(), A.data_ptr(), arr, len(layer.Bs), layer.C.data_ptr(), D.data_ptr(), s1.data_ptr(), layer.s2.data_ptr(), None,
                          M, layer.N, layer.K, layer.ws.data_ptr(), -1, 0, ctypes.c_void_p(st), 16, None, n, out)
    assert rc == 0
    return np.array(out[:]) * 1e3
res = {k: [] for k in cases}
for r in range(4):
    for k, A in cases.items():
        run(A, 2)
        res[k].extend(run(A))
for k, v in res.items():
    print(f"{k:28s} median {np.median(v):7.1f} us   min {np.min(v):7.1f}")
wz = [torch.zeros_like(b) for b in layer.Bs]
arr = (ctypes.c_void_p * len(wz))(*[b.data_ptr() for b in wz])
v = []
for r in range(4):
    run(A0, 2); v.extend(run(A0))
print(f"{'gaussian int8, weights = 0':28s} median {np.median(v):7.1f} us   min {np.min(v):7.1f}")
