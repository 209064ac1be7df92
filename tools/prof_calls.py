#!/usr/bin/env python3
"""Run a few qqq_gemm configurations back to back (for rocprofv3 --kernel-trace --stats)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="1,16,128,1024,4096")
ap.add_argument("--mode", default="pc", help="pc | g128 | g128x (per-group with the opt-in expanded int8 weights)")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--tune", default="{}")
ap.add_argument("--nk", default=f"{Bn.N_FULL},{Bn.K_FULL}")
args = ap.parse_args()
dev = torch.device("cuda:0")
NN, KK = [int(x) for x in args.nk.split(",")]
layer = Bn.Layer(dev, grouped=args.mode.startswith("g128"), N=NN, K=KK, expand=(args.mode == "g128x"))
tune = json.loads(args.tune) or None
for M in [int(x) for x in args.ms.split(",")]:
    A, s1 = Bn.make_tokens(dev, M, M, K=KK)
    D = torch.empty((M, NN), dtype=torch.float16, device=dev)
    layer.time_calls(A, s1, D, 3, tune=tune)
    t = layer.time_calls(A, s1, D, args.iters, tune=tune) * 1e3
    print(f"M={M} mean {t.mean():.1f} us  min {t.min():.1f}  tune={tune}")
