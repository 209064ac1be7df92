#!/usr/bin/env python3
"""Large-m comparison at the BASELINE layer: tiled (auto) vs the panel kernel's shapes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
dev = torch.device("cuda:0")
modes = os.environ.get("MODES", "pc").split(",")
Ms = [int(x) for x in os.environ.get("MS", "256,512,1024,2048,4096").split(",")]
tunes = eval(os.environ.get("TUNES", "[dict(kernel=2), dict(kernel=4,mt=8,bm=128), dict(kernel=4,mt=8,bm=256,pf=3), dict(kernel=4,mt=16), dict(kernel=4,mt=16,ksplit=1)]"))
for mode in modes:
    layer = Bn.Layer(dev, grouped=(mode == "g128"), nbuf=4)
    for M in Ms:
        A, s1 = Bn.make_tokens(dev, M, M)
        D = torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev)
        for tune in tunes:
            try:
                layer.time_calls(A, s1, D, 2, tune=tune)
                v = layer.time_calls(A, s1, D, 8, tune=tune) * 1e3
                print(f"{mode} M={M:5d} {np.median(v):8.1f} us (min {v.min():8.1f})  {Bn.algorithmic_ops(M, Bn.N_FULL, Bn.K_FULL)/np.median(v)/1e6:7.0f} TOPS  {tune}")
            except Exception as e:
                print(f"{mode} M={M} {tune} ERR {e}")
            sys.stdout.flush()
