#!/usr/bin/env python3
"""Phase timeline of the panel kernel from inside the kernel (100 MHz wall clock, thread 0 of every workgroup).

Build (where hipcc is):   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DQQQ_PANEL_TRACE \
                                -o qqq_amd/libtrace.so qqq_amd/csrc/qqq_w4a8.hip
Run (GPU box):            QQQ_AMD_LIB=qqq_amd/libtrace.so MS=128 python tools/trace_panel.py
Clock points: 0 entry, 1 prologue done (first operands unpacked), 2 main loop done, 3 k-groups met in LDS, 4 ticket known,
5 deposit in memory (depositors) / every deposit seen (finisher), 6 fold done, 7 epilogue stores drained."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import _lib

L = _lib.lib()
L.qqq_trace_set.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
grouped = os.environ.get("MODE", "pc") == "g128"
tune = eval(os.environ.get("TUNE", "None"))
NN, KK = [int(x) for x in os.environ.get("NK", f"{Bn.N_FULL},{Bn.K_FULL}").split(",")]
layer = Bn.Layer(dev, grouped=grouped, nbuf=4, N=NN, K=KK)
NAMES = ["entry", "prologue", "main loop", "lds meet", "ticket", "deposit/spin", "fold", "epilogue"]


def q(v):
    v = np.asarray(v, dtype=np.float64)
    return f"{v.min():6.2f} {np.median(v):6.2f} {v.max():6.2f}"


for M in [int(x) for x in os.environ.get("MS", "128").split(",")]:
    A, s1 = Bn.make_tokens(dev, M, M, K=KK)
    D = torch.empty((M, NN), dtype=torch.float16, device=dev)
    p = _lib.plan(M, NN, KK, 128 if grouped else -1, Bn.MAX_PAR, tune=tune)
    assert p["kernel"] == 4, p
    rows, bn, ks = 16 * p["mt"], p["bm"], p["ksplit"]
    nwg = -(-NN // bn) * ks * -(-M // rows)
    buf = torch.zeros((nwg, 16), dtype=torch.int64, device=dev)
    L.qqq_trace_set(None)
    ev = layer.time_calls(A, s1, D, 8, tune=tune)
    print(f"== M={M} {'g128' if grouped else 'per-channel'}  plan {p}\n   {nwg} workgroups; event-timed (trace stores compiled in, buffer off): median {np.median(ev) * 1e3:.1f} us")
    runs = []
    for rep in range(6):
        buf.zero_()
        assert L.qqq_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
        layer.time_calls(A, s1, D, 1, tune=tune, rotate=False) if rep == 0 else layer.time_calls(A, s1, D, 1, tune=tune)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.int64)
        L.qqq_trace_set(None)
        runs.append(t)
    t = runs[-1]
    T0 = t[:, 0].min()
    us = (t[:, :8] - T0) / 100.0  # 100 MHz
    tick, xcc = t[:, 8], t[:, 9]
    last = tick == ks - 1 if ks > 1 else np.ones(nwg, bool)
    print(f"   clock points, us after the first workgroup's entry (min median max over workgroups); kernel end = {us[:, 7].max():.2f}")
    for who, sel in (("finishers", last), ("depositors", ~last)):
        if sel.sum() == 0:
            continue
        print(f"   -- {who} ({int(sel.sum())})")
        for i in range(8):
            col = us[sel, i]
            if (t[sel, i] == 0).all():
                continue
            dur = "" if i == 0 else f"   phase {q(us[sel, i] - us[sel, i - 1])}"
            print(f"      {i} {NAMES[i]:14s} at {q(col)}{dur}")
    ends = [((r[:, 7].max() - r[:, 0].min()) / 100.0) for r in runs]
    print(f"   first entry -> last store drained, 6 launches: {' '.join(f'{e:.1f}' for e in ends)} us")
    if ks > 1:
        # do the slices of a tile share an XCD?
        g = xcc.reshape(-1, ks, -(-NN // bn))  # (mblk, slice, strip)
        same = (g == g[:, :1, :]).all(axis=1).mean()
        print(f"   tiles whose {ks} slices ran on one XCD: {same * 100:.0f} %;  entry skew between a tile's slices (us): {q((us[:, 0].reshape(g.shape).max(axis=1) - us[:, 0].reshape(g.shape).min(axis=1)).ravel())}")
        sl = np.broadcast_to(np.arange(ks)[None, :, None], g.shape).ravel()  # the K slice of every workgroup (grid: strip fastest, then slice, then m-block)
        print(f"   the finisher (last arrival) is the LAST K slice -- the longer one when the slices are uneven (plan skew {p.get('skew', 0)}) -- in {100 * (sl[last] == ks - 1).mean():.0f} % of the tiles")
        fin_wait = us[last, 5] - us[last, 4]
        print(f"   finisher: wait for the other slices' deposits {q(fin_wait)} us, fold {q(us[last, 6] - us[last, 5])} us, epilogue {q(us[last, 7] - us[last, 6])} us")
