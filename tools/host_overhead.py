#!/usr/bin/env python3
"""Host-side cost of one qqq_gemm call (eager, M=1 decode: the kernel takes ~17 us): torch custom-op path vs the direct
ctypes path vs the native back-to-back loop (qqq_dev_bench_gemm)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import ops
dev = torch.device("cuda:0")
layer = Bn.Layer(dev, grouped=False, nbuf=2, N=4096, K=4096)
A, s1 = Bn.make_tokens(dev, 1, 1, K=4096)
D = torch.empty((1, 4096), dtype=torch.float16, device=dev)
def loop(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()          # enqueue time only
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
f_op = lambda: ops.qqq_gemm(A, layer.Bs[0], layer.C, D, s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, 16)
f_direct = lambda: ops._qqq_gemm_impl(A, layer.Bs[0], layer.C, D, s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, 16)
f_disp = lambda: ops._qqq_gemm_op(A, layer.Bs[0], layer.C, D, s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, 16)
f_native = lambda: torch.ops.qqq_amd_native.qqq_gemm(A, layer.Bs[0], layer.C, D, s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, 16)
print("compiled torch binding present:", ops._ext() is not None)
print("qqq_gemm (eager: compiled binding if present)   : enqueue %.1f us/call, wall %.1f us/call" % loop(f_op))
print("ctypes binding (_qqq_gemm_impl)                 : enqueue %.1f us/call, wall %.1f us/call" % loop(f_direct))
print("python custom op through the dispatcher         : enqueue %.1f us/call, wall %.1f us/call" % loop(f_disp))
if ops._ext() is not None:
    print("TORCH_LIBRARY op qqq_amd_native::qqq_gemm       : enqueue %.1f us/call, wall %.1f us/call" % loop(f_native))
import numpy as np
print("native loop: %.1f us/call (HIP events)" % float(np.mean(layer.time_calls(A, s1, D, 200)) * 1e3))
x = torch.randn((1, 4096), device=dev, dtype=torch.float16)
print("dynamic_quant custom op: enqueue %.1f us/call, wall %.1f us/call" % loop(lambda: ops.dynamic_quant(x)))
from qqq_amd import QuantLinear
ql = QuantLinear(4, -1, 4096, 4096, bias=False).to(dev)
ql.B.copy_(layer.Bs[0]); ql.s_channel.copy_(layer.s2)
print("QuantLinear.forward (one binding call): enqueue %.1f us/call, wall %.1f us/call" % loop(lambda: ql(x)))
