#!/usr/bin/env python3
"""Bit-for-bit check of tuning variants against the automatic dispatch at the BASELINE layer size (or --nk), both modes.
usage: python tools/check_variant.py --ms 4096,1000 --tunes "[dict(kernel=5), dict(kernel=5, pf=3)]" [--modes pc,g128] [--ref "dict(kernel=2)"]"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn
from gpu_util import GemmHarness

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="4096")
ap.add_argument("--modes", default="pc,g128")
ap.add_argument("--tunes", default="[dict(kernel=5)]")
ap.add_argument("--ref", default="None")
ap.add_argument("--nk", default=f"{Bn.N_FULL},{Bn.K_FULL}")
args = ap.parse_args()
dev = torch.device("cuda:0")
NN, KK = [int(x) for x in args.nk.split(",")]
bad = 0
for mode in args.modes.split(","):
    grouped = mode == "g128"
    layer = Bn.Layer(dev, grouped=grouped, nbuf=1, N=NN, K=KK)
    h = GemmHarness(layer.Bs[0], layer.s2, layer.s3 if grouped else None, dev)
    for M in [int(x) for x in args.ms.split(",")]:
        A, s1 = Bn.make_tokens(dev, M, M, K=KK)
        D0, a0 = h.run(A, s1, eval(args.ref))
        for tune in eval(args.tunes):
            D, a = h.run(A, s1, tune)
            ok = np.array_equal(a, a0) and np.array_equal(D.view(np.uint16), D0.view(np.uint16))
            nbad = int((a != a0).sum())
            print(f"{mode} M={M} {tune}: {'bit-exact' if ok else 'MISMATCH'} acc_diff={nbad}" + ("" if ok else f" first rows {np.unique(np.nonzero(a != a0)[0])[:8]} cols {np.unique(np.nonzero(a != a0)[1])[:8]}"), flush=True)
            bad += not ok
    del layer, h
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
