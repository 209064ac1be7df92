#!/usr/bin/env python3
"""Phase timeline of the wide kernel from inside the kernel (thread 0 of every workgroup: 100 MHz wall clock at entry / loop
start / loop end / last store drained, shader clock at the top of the first 12 trips of the 4-stage loop).

Build (where hipcc is):   QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_trace.so QQQ_AMD_CXXFLAGS=-DQQQ_PANEL_TRACE python -m qqq_amd.build
Run (GPU box):            QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_trace.so MS=4096 python tools/trace_wide.py"""
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
tune = eval(os.environ.get("TUNE", "dict(kernel=5)"))
NN, KK = [int(x) for x in os.environ.get("NK", f"{Bn.N_FULL},{Bn.K_FULL}").split(",")]
layer = Bn.Layer(dev, grouped=grouped, nbuf=4, N=NN, K=KK)
q = lambda v: f"{np.min(v):8.2f} {np.median(v):8.2f} {np.max(v):8.2f}"
for M in [int(x) for x in os.environ.get("MS", "4096").split(",")]:
    A, s1 = Bn.make_tokens(dev, M, M, K=KK)
    D = torch.empty((M, NN), dtype=torch.float16, device=dev)
    pl = _lib.plan(M, NN, KK, 128 if grouped else -1, 16, tune=tune)  # the grid of the shape that will run (tile height / width, K slices)
    nwg = -(-NN // pl["bm"]) * -(-M // (16 * pl["mt"])) * pl["ksplit"]
    buf = torch.zeros((nwg, 16), dtype=torch.int64, device=dev)
    L.qqq_trace_set(None)
    ev = layer.time_calls(A, s1, D, 6, tune=tune)
    print(f"== M={M} N={NN} K={KK} {'g128' if grouped else 'per-channel'} tune={tune}: {nwg} workgroups, event-timed median {np.median(ev) * 1e3:.1f} us")
    for rep in range(3):
        buf.zero_()
        assert L.qqq_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
        layer.time_calls(A, s1, D, 1, tune=tune)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.int64)
        L.qqq_trace_set(None)
    T0 = t[:, 0].min()
    us = (t[:, :4] - T0) / 100.0
    first = us[:, 0] < 5.0  # workgroups of the first round
    for name, sel in (("first round", first), ("later rounds", ~first)):
        if sel.sum() == 0:
            continue
        print(f"   -- {name} ({int(sel.sum())} workgroups)   min / median / max, us")
        print(f"      entry at            {q(us[sel, 0])}")
        print(f"      prologue            {q(us[sel, 1] - us[sel, 0])}")
        print(f"      main loop           {q(us[sel, 2] - us[sel, 1])}   per 128-k stage {np.median(us[sel, 2] - us[sel, 1]) / ((KK // 64 + 1) // 2):.3f}")
        print(f"      epilogue            {q(us[sel, 3] - us[sel, 2])}")
    if pl["ksplit"] > 1 and (t[:, 4] != 0).any():  # split K (plain trace build): ticket / deposit published / deposits complete, per role
        u = (t[:, :7] - T0) / 100.0
        arr, ex = t[:, 7] >> 1, t[:, 7] & 1
        print(f"   split K: {int(ex.sum())} of {nwg} workgroups exchange; per role, us after the first entry (min / median / max)")
        for name, sel in (("depositors (classic)", (ex == 0) & (arr < pl['ksplit'] - 1)), ("finishers (classic)", (ex == 0) & (arr == pl['ksplit'] - 1)), ("exchanging", ex == 1)):
            if sel.sum() == 0:
                continue
            print(f"      {name}: {int(sel.sum())}")
            print(f"         loop end          {q(u[sel, 2])}")
            print(f"         ticket - loop end {q(u[sel, 4] - u[sel, 2])}")
            if (t[sel, 5] != 0).any():
                print(f"         deposit published - ticket   {q(u[sel, 5] - u[sel, 4])}")
            if (t[sel, 6] != 0).any():
                print(f"         deposits complete - ticket   {q(u[sel, 6] - u[sel, 4])}")
                print(f"         end - deposits complete      {q(u[sel, 3] - u[sel, 6])}")
            print(f"         end - loop end    {q(u[sel, 3] - u[sel, 2])}")
    trips = np.diff(t[:, 4:16], axis=1).astype(np.float64)
    ok = (t[:, 4:16] != 0).all(axis=1) & (pl["ksplit"] == 1)
    if ok.any():
        print(f"   shader clocks per 3-stage trip (384 MFMAs x 16 = 6144 matrix-pipe clocks), trips 1..11: median over workgroups " + " ".join(f"{np.median(trips[ok, j]):.0f}" for j in range(trips.shape[1])))
        print(f"      min over workgroups " + " ".join(f"{np.min(trips[ok, j]):.0f}" for j in range(trips.shape[1])))
    print(f"   kernel end {us[:, 3].max():.1f} us after the first entry")
