#!/usr/bin/env python3
"""Decode / small-m regime: column kernel (per-lane activation loads vs LDS-shared activations) vs stream kernel + reduce."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "8192x21760,4096x4096,11008x4096,4096x11008").split(",")]
Ms = [int(x) for x in os.environ.get("MS", "1,8,12,16,24,32").split(",")]
for (N, K) in shapes:
    for mode in os.environ.get("MODES", "pc,g128").split(","):
        layer = Bn.Layer(dev, grouped=(mode == "g128"), nbuf=4 if N * K > 6e7 else 12, N=N, K=K)
        for M in Ms:
            A, s1 = Bn.make_tokens(dev, M, M, K=K)
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            def t(tune):
                layer.time_calls(A, s1, D, 3, tune=tune)
                v = layer.time_calls(A, s1, D, 16, tune=tune) * 1e3
                return f"{np.median(v):6.1f}/{v.min():5.1f}"
            mt = 1 if M <= 16 else 2
            print(f"N={N:5d} K={K:5d} {mode:4s} M={M:3d} auto {t(None)} | stream {t(dict(kernel=1))} | col-lane {t(dict(kernel=3, glds=2, mt=mt))} | col-lds pf3 {t(dict(kernel=3, glds=1, mt=mt))} pf2 {t(dict(kernel=3, glds=1, mt=mt, pf=2))} pf4 {t(dict(kernel=3, glds=1, mt=mt, pf=4))}")
            sys.stdout.flush()
        del layer
        torch.cuda.empty_cache()
