#!/usr/bin/env python3
"""What a ragged token count costs, and what splitting the call along M would save (rows are independent): one call of M tokens against two calls --
the largest multiple of 256 below M, then the remainder -- on the same stream, event-timed as a pair, rotating cold weights.
usage: NK=8192,21760 MS=4097,4100,4224 python tools/ragged_m.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import ops

dev = torch.device("cuda:0")
NN, KK = [int(x) for x in os.environ.get("NK", f"{Bn.N_FULL},{Bn.K_FULL}").split(",")]
Ms = [int(x) for x in os.environ.get("MS", "4097,4100,4224").split(",")]
grouped = os.environ.get("MODE", "pc") == "g128"
step = int(os.environ.get("STEP", "256"))
layer = Bn.Layer(dev, grouped=grouped, nbuf=int(os.environ.get("NBUF", "5")), N=NN, K=KK)
reps = int(os.environ.get("REPS", "24"))


def gemm(A, s1, D, j):
    ops.qqq_gemm(A, layer.Bs[j % len(layer.Bs)], layer.C, D, s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, Bn.MAX_PAR)


def timed(fn):
    for j in range(3):
        fn(j)
    torch.cuda.synchronize()
    out = []
    for j in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(j); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(out))


for M in Ms:
    A, s1 = Bn.make_tokens(dev, M, M, K=KK)
    D = torch.empty((M, NN), dtype=torch.float16, device=dev)
    M0 = (M // step) * step
    whole = timed(lambda j: gemm(A, s1, D, j))
    line = f"N={NN} K={KK} {'g128' if grouped else 'pc'} M={M}: one call {whole:8.1f} us"
    if 0 < M0 < M:
        def two(j):
            gemm(A[:M0], s1[:M0], D[:M0], j)
            gemm(A[M0:], s1[M0:], D[M0:], j)
        split = timed(two)
        base = timed(lambda j: gemm(A[:M0], s1[:M0], D[:M0], j))
        line += f" | {M0} + {M - M0} tokens in two calls {split:8.1f} us ({100 * (whole / split - 1):+.0f} %) | {M0} tokens alone {base:8.1f} us"
    print(line, flush=True)
