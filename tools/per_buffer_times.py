#!/usr/bin/env python3
"""Per-call times of one sweep point grouped by the rotating weight copy it read: is the spread of `per_m` (1024 tokens: median 10 % above min) a property of some copies (placement) or of time?
usage: MS=1024,4096 CALLS=120 python tools/per_buffer_times.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
dev = torch.device("cuda:0")
nb = int(os.environ.get("NBUF", "12"))
layer = Bn.Layer(dev, grouped=False, nbuf=nb)
print("weight copies at (MiB offsets from the first):", [round((b.data_ptr() - layer.Bs[0].data_ptr()) / 2**20, 2) for b in layer.Bs])
for M in [int(x) for x in os.environ.get("MS", "1024,4096").split(",")]:
    A, s1 = Bn.make_tokens(dev, M, M)
    D = torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev)
    layer.time_calls(A, s1, D, 3)
    layer._rot = 0
    n = int(os.environ.get("CALLS", "120"))
    t = layer.time_calls(A, s1, D, n) * 1e3
    by = [np.median(t[i::nb]) for i in range(nb)]
    print(f"M={M}: median {np.median(t):.1f} min {t.min():.1f} max {t.max():.1f}; median per weight copy: " + " ".join(f"{v:.1f}" for v in by))
    print("   in call order:", " ".join(f"{v:.0f}" for v in t[:48]))
