#!/usr/bin/env python3
"""Fixed cost vs per-stage cost of the panel kernel: same launch geometry, K scaled (M, N fixed)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "128"))
tunes = eval(os.environ.get("TUNES", "[dict(kernel=4,mt=8,bm=128,waves=8,ksplit=4,pf=4), dict(kernel=4,mt=8,bm=128,waves=8,ksplit=2,pf=4), dict(kernel=4,mt=8,bm=128,waves=8,ksplit=1,pf=4), dict(kernel=1)]"))
for K in (2560, 5120, 10240, 20480):
    layer = Bn.Layer(dev, grouped=False, nbuf=4, K=K)
    A, s1 = Bn.make_tokens(dev, M, M, K=K)
    D = torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev)
    for tune in tunes:
        layer.time_calls(A, s1, D, 3, tune=tune)
        v = layer.time_calls(A, s1, D, 16, tune=tune) * 1e3
        print(f"K={K:6d} stages/slice={K//128//max(tune.get('ksplit',1),1):4d} {np.median(v):7.1f} us (min {v.min():6.1f})  {tune}")
    del layer
    torch.cuda.empty_cache()
