#!/usr/bin/env python3
"""Seam timeline of the persistent tile walk (wide kernel, tune glds=2) from inside the kernel: thread 0 of every workgroup writes
the 100 MHz wall clock at entry, at the start of its stage loop, and at the start / end of the first seven tile seams.

Build (where hipcc is):   QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_trace.so QQQ_AMD_CXXFLAGS=-DQQQ_PANEL_TRACE python -m qqq_amd.build
Run (GPU box):            QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_trace.so NK=4096,4096 MS=8192 python tools/trace_chain.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import _lib

L = _lib.lib()
L.qqq_trace_set.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
grouped = os.environ.get("MODE", "pc") == "g128"
tune = eval(os.environ.get("TUNE", "dict(kernel=5,glds=2)"))
NN, KK = [int(x) for x in os.environ.get("NK", "4096,4096").split(",")]
layer = Bn.Layer(dev, grouped=grouped, nbuf=4, N=NN, K=KK)
q = lambda v: f"{np.min(v):8.2f} {np.median(v):8.2f} {np.max(v):8.2f}"
for M in [int(x) for x in os.environ.get("MS", "8192").split(",")]:
    A, s1 = Bn.make_tokens(dev, M, M, K=KK)
    D = torch.empty((M, NN), dtype=torch.float16, device=dev)
    pl = _lib.plan(M, NN, KK, 128 if grouped else -1, 16, tune=tune)
    tiles = -(-NN // pl["bm"]) * -(-M // (16 * pl["mt"]))
    nwg = 256
    buf = torch.zeros((nwg, 16), dtype=torch.int64, device=dev)
    L.qqq_trace_set(None)
    ev = layer.time_calls(A, s1, D, 6, tune=tune)
    print(f"== M={M} N={NN} K={KK} {'g128' if grouped else 'per-channel'} tune={tune} plan={pl}: {tiles} tiles on {nwg} workgroups, event-timed median {np.median(ev) * 1e3:.1f} us")
    for rep in range(3):
        buf.zero_()
        assert L.qqq_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
        layer.time_calls(A, s1, D, 1, tune=tune)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.int64)
        L.qqq_trace_set(None)
    T0 = t[:, 0].min()
    us = np.where(t != 0, (t - T0) / 100.0, np.nan)
    print(f"      entry at                  {q(us[:, 0])}")
    print(f"      prologue                  {q(us[:, 1] - us[:, 0])}")
    prev = us[:, 1]
    nst = KK // 128
    for j in range(7):
        a, b = us[:, 2 + 2 * j], us[:, 3 + 2 * j]
        ok = ~np.isnan(a) & ~np.isnan(b)
        if not ok.any():
            break
        print(f"      tile {j}: {int(ok.sum()):3d} workgroups  loop {q((a - prev)[ok])}  ({np.median((a - prev)[ok]) / nst:.3f} us per stage)   flush {q((b - a)[ok])}   seam starts at {q(a[ok])}")
        prev = b
    print(f"      last stamp {np.nanmax(us):.1f} us after the first entry")
