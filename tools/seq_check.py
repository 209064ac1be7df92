#!/usr/bin/env python3
"""Does a sweep point cost the same inside the bench step's sequence (1, 16, 128, 1024, 4096 tokens back to back, rotating weight copies) as in a loop of its own?
Event-timed per call (bench.Layer.time_calls), per-channel, N = 8192, K = 21760.  ORDER=... changes the sequence."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
dev = torch.device("cuda:0")
layer = Bn.Layer(dev, grouped=False, nbuf=12)
order = [int(x) for x in os.environ.get("ORDER", "1,16,128,1024,4096").split(",")]
toks = {M: Bn.make_tokens(dev, M, M) for M in set(order)}
D = {M: torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev) for M in set(order)}
for M in set(order):
    layer.time_calls(*toks[M], D[M], 3)
alone = {M: layer.time_calls(*toks[M], D[M], 60) * 1e3 for M in sorted(set(order))}
seq = {M: [] for M in set(order)}
for r in range(60):
    for M in order:
        seq[M].append(float(layer.time_calls(*toks[M], D[M], 1)[0]) * 1e3)
print(f"# order {order}; per call us: loop of its own (60 calls) median / min   |   inside the sequence (60 rounds) median / min")
for M in sorted(set(order)):
    print(f"M={M:5d}  alone {np.median(alone[M]):7.1f} {np.min(alone[M]):7.1f}   in sequence {np.median(seq[M]):7.1f} {np.min(seq[M]):7.1f}")
