#!/usr/bin/env python3
"""Interleaved A/B timing of library builds x tuning variants (within one process, rotating weights).
usage: LIBS="a.so,b.so" TUNES="[dict(...),...]" MS="4096" MODE=pc python tools/ab.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import _dev, _lib

DEV = _dev.lib()

def load(path):
    """an operator-library build (possibly a tuning variant); timed through the dev library's event loop"""
    L = ctypes.CDLL(path)
    L.qqq_amd_last_error.restype = ctypes.c_char_p
    return L  # (gemm_ex2_ptr / gemm_ex_ptr only take the symbol's address)

dev = torch.device("cuda:0")
libs = [p for p in os.environ.get("LIBS", "qqq_amd/libqqq_amd.so").split(",")]
Ls = [load(os.path.join(ROOT, p) if not os.path.isabs(p) else p) for p in libs]
tunes = eval(os.environ.get("TUNES", "[None]"))
Ms = [int(x) for x in os.environ.get("MS", "4096").split(",")]
grouped = os.environ.get("MODE", "pc") == "g128"
rounds = int(os.environ.get("ROUNDS", "5"))
iters = int(os.environ.get("ITERS", "4"))
NN, KK = [int(x) for x in os.environ.get("NK", f"{Bn.N_FULL},{Bn.K_FULL}").split(",")]
# W8=1 (per-group): every weight copy also as expanded int8 (qqq_expand_int8), handed to every call -- a variant with tune w8=-1 ignores them
layer = Bn.Layer(dev, grouped=grouped, nbuf=int(os.environ.get("NBUF", "4")) or Bn.copies_for(NN, KK), N=NN, K=KK, expand=os.environ.get("W8") == "1")
rot = [0]  # the weight copies rotate on from one timed group to the next (NBUF=0: as many copies as make the rotation 1.1 GB long)
for M in Ms:
    A, s1 = Bn.make_tokens(dev, M, M, K=KK)
    D = torch.empty((M, NN), dtype=torch.float16, device=dev)
    if os.environ.get('ZERO') == '1':
        A.zero_()
        for b in layer.Bs: b.zero_()
    res = {}
    def run(L, tune, n):
        out = (ctypes.c_float * n)()
        tn = None
        if tune:
            tn = _lib.QQQTune()
            for k, v in tune.items(): setattr(tn, k, int(v))
        st = torch.cuda.current_stream(dev).cuda_stream
        nb = len(layer.Bs)
        r0 = rot[0]
        arr = (ctypes.c_void_p * nb)(*[layer.Bs[(r0 + i) % nb].data_ptr() for i in range(nb)])
        rot[0] = (rot[0] + n) % nb
        if layer.W8s is not None:
            arr8 = (ctypes.c_void_p * nb)(*[layer.W8s[(r0 + i) % nb].data_ptr() for i in range(nb)])
            rc = DEV.qqq_dev_bench_gemm2(_dev.gemm_ex2_ptr(L), A.data_ptr(), arr, arr8, len(layer.Bs), layer.C.data_ptr(), D.data_ptr(), s1.data_ptr(), layer.s2.data_ptr(),
                                   layer.s3.data_ptr() if layer.s3.numel() else None, M, layer.N, layer.K, layer.ws.data_ptr(),
                                   layer.groupsize, 0, ctypes.c_void_p(st), 16, ctypes.byref(tn) if tn is not None else None, n, out)
        else:
            rc = DEV.qqq_dev_bench_gemm(_dev.gemm_ex_ptr(L), A.data_ptr(), arr, len(layer.Bs), layer.C.data_ptr(), D.data_ptr(), s1.data_ptr(), layer.s2.data_ptr(),
                                  layer.s3.data_ptr() if layer.s3.numel() else None, M, layer.N, layer.K, layer.ws.data_ptr(),
                                  layer.groupsize, 0, ctypes.c_void_p(st), 16, ctypes.byref(tn) if tn is not None else None, n, out)
        assert rc == 0, (rc, L.qqq_amd_last_error())
        return np.array(out[:]) * 1e3
    # FLUSH=1 (tools/visits/r5_v23.sh): 640 MB of unrelated traffic (read + write) in front of every timed group -- with ITERS = NBUF every call of the group then
    # meets a cold copy whatever was timed before.  (Its dirty lines are written back under the first launches of the group: take groups of >= 8 calls.)
    if os.environ.get("FLUSH") == "1":
        junk = torch.zeros(640 << 20, dtype=torch.uint8, device=dev)
        _run = run
        def run(L, tune, n):
            junk.add_(1)
            return _run(L, tune, n)
    for li, L in enumerate(Ls):
        for tune in tunes: run(L, tune, 2)
    for r in range(rounds):
        for li, L in enumerate(Ls):
            for ti, tune in enumerate(tunes):
                res.setdefault((li, ti), []).extend(run(L, tune, iters))
    ops = Bn.algorithmic_ops(M, NN, KK); byts = Bn.algorithmic_bytes(M, NN, KK, grouped)
    for (li, ti), v in sorted(res.items()):
        v = np.array(v)
        print(f"M={M} {os.path.basename(libs[li]):24s} {str(tunes[ti]):60s} med {np.median(v):8.1f} us  min {v.min():8.1f}  {ops/np.median(v)/1e6:7.0f} TOPS {byts/np.median(v)/1e3:6.0f} GB/s")
