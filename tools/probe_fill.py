#!/usr/bin/env python3
"""Per-CU read bandwidth of the MI355X as seen by one workgroup per CU (qqq_dev_probe_fill): L2-resident window
shared by all workgroups (L2 -> L1 fill) and disjoint HBM windows, for 1 / 64 / 256 workgroups."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qqq_amd import _dev
L = _dev.lib()
dev = torch.device("cuda:0")
buf = torch.randint(0, 2**31 - 1, ((512 << 20) // 4,), dtype=torch.int32, device=dev)  # 512 MB
sink = torch.zeros(4, dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def run(stride, per_wg, nwg, reps, unroll):
    ms = ctypes.c_float()
    rc = L.qqq_dev_probe_fill(buf.data_ptr(), stride, per_wg, nwg, reps, unroll, sink.data_ptr(), 0, st, ctypes.byref(ms))
    assert rc == 0
    tot = per_wg * reps * nwg
    return tot / (ms.value * 1e-3) / 1e9, ms.value * 1e3
print("mode                      nwg  unroll   GB/s total   GB/s per WG    us")
for unroll in (2, 8):
    for nwg in (1, 16, 64, 256):
        g, us = run(0, 1 << 20, nwg, 64, unroll)          # shared 1 MB window, 64 passes: L2 hits
        print(f"L2-resident shared 1MB   {nwg:4d}  {unroll:5d}   {g:10.0f}   {g/nwg:10.1f}   {us:8.1f}")
    for nwg in (1, 16, 64, 256):
        g, us = run(2 << 20, 2 << 20, nwg, 1, unroll)       # disjoint 2 MB windows (512 MB total at 256): HBM
        print(f"HBM disjoint 2MB/WG      {nwg:4d}  {unroll:5d}   {g:10.0f}   {g/nwg:10.1f}   {us:8.1f}")
