#!/usr/bin/env python3
"""Decode regime, cold: the W4A8 GEMM (automatic dispatch, both modes) against torch's fp16 GEMM on the same GPU for the linear shapes of Llama-2-7B / 13B / 70B,
1 ... 16 tokens.  Every call meets weights no cache holds: 1.1 GB of rotating copies per layer for the int4 weights (bench.copies_for) and for the fp16 ones.
    SHAPES=4096x4096,... MS=1,4,8,16 python tools/decode_table.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import _lib

dev = torch.device("cuda:0")
DEFAULT = "4096x4096,12288x4096,11008x4096,22016x4096,4096x11008,5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672"
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", DEFAULT).split(",")]
Ms = [int(x) for x in os.environ.get("MS", "1,4,8,16").split(",")]


def fp16_us(N, K, M, iters=24):
    n = int(min(64, max(3, -(-1.1e9 // (N * K * 2)))))
    Ws = [torch.randn((K, N), device=dev, dtype=torch.float16) * 0.02 for _ in range(n)]
    x = torch.randn((M, K), device=dev, dtype=torch.float16)
    for i in range(n):
        torch.matmul(x, Ws[i])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(ev):
        a.record(); torch.matmul(x, Ws[i % n]); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]) * 1e3)


print("# N x K (out x in), tokens: W4A8 per-channel / per-group us (kernel family), fp16 torch.matmul us, speed-up per-channel / per-group; all cold")
for (N, K) in shapes:
    layers = {g: Bn.Layer(dev, grouped=g, nbuf=Bn.copies_for(N, K), N=N, K=K) for g in (False, True)}
    for M in Ms:
        A, s1 = Bn.make_tokens(dev, M, M, K=K)
        D = torch.empty((M, N), dtype=torch.float16, device=dev)
        us = {}
        for g, layer in layers.items():
            layer.time_calls(A, s1, D, 4)
            us[g] = float(np.median(layer.time_calls(A, s1, D, 24) * 1e3))
        f = fp16_us(N, K, M)
        fam = [Bn.FAMILY[_lib.plan(M, N, K, gs, 16)["kernel"]] for gs in (-1, 128)]
        print(f"N={N:5d} K={K:5d} M={M:3d}  W4A8 {us[False]:6.1f} ({fam[0]:6s}) / {us[True]:6.1f} ({fam[1]:6s})   fp16 {f:6.1f}   x{f / us[False]:.2f} / x{f / us[True]:.2f}")
        sys.stdout.flush()
    del layers
    torch.cuda.empty_cache()
