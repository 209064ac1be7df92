#!/usr/bin/env python3
"""Where does a tuning variant differ from the reference variant?  Mismatch map by tile / row-in-tile / column-in-tile (debugging aid).
usage: python tools/diag_chain.py --nk 4096,4096 --m 8192 --mode pc --tune "dict(kernel=5,glds=2)" """
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn
from gpu_util import GemmHarness
ap = argparse.ArgumentParser()
ap.add_argument("--nk", default="4096,4096"); ap.add_argument("--m", type=int, default=8192); ap.add_argument("--mode", default="pc")
ap.add_argument("--tune", default="dict(kernel=5,glds=2)"); ap.add_argument("--ref", default="dict(kernel=2)")
ap.add_argument("--rows", type=int, default=256); ap.add_argument("--cols", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, K = [int(x) for x in args.nk.split(",")]
grouped = args.mode == "g128"
layer = Bn.Layer(dev, grouped=grouped, nbuf=1, N=N, K=K)
h = GemmHarness(layer.Bs[0], layer.s2, layer.s3 if grouped else None, dev)
A, s1 = Bn.make_tokens(dev, args.m, args.m, K=K)
D0, a0 = h.run(A, s1, eval(args.ref))
D, a = h.run(A, s1, eval(args.tune))
bad = a != a0
print(f"{args.mode} M={args.m} N={N} K={K} {args.tune}: acc mismatches {int(bad.sum())} of {bad.size} ({bad.mean():.4f}); D mismatches {int((D.view(np.uint16) != D0.view(np.uint16)).sum())}")
R, C = args.rows, args.cols
tm, tn = -(-args.m // R), -(-N // C)
frac = np.zeros((tm, tn))
for i in range(tm):
    for j in range(tn):
        frac[i, j] = bad[i * R:(i + 1) * R, j * C:(j + 1) * C].mean()
np.set_printoptions(linewidth=250, precision=2, suppress=True)
print("per-tile mismatch fraction (rows = m-tiles, cols = n-tiles):"); print(frac[:, :min(tn, 32)])
rows = np.zeros(R); cols = np.zeros(C); cnt = 0
for i in range(tm):
    for j in range(tn):
        t = bad[i * R:(i + 1) * R, j * C:(j + 1) * C]
        if t.shape == (R, C) and t.any():
            rows += t.mean(axis=1); cols += t.mean(axis=0); cnt += 1
if cnt:
    print("mismatch fraction by row-in-tile, 16 per line (averaged over tiles with any mismatch):"); print((rows / cnt).reshape(-1, 16))
    print("by column-in-tile, 16 per line:"); print((cols / cnt).reshape(-1, 16))
    i, j = np.argwhere(bad)[0]
    print("first mismatch at", (i, j), "expected", a0[i, j:j + 8], "got", a[i, j:j + 8])
    # does the wrong value equal the right value of some other tile / a partial sum?
    print("row", i, "expected[:8]", a0[i, :8], "got", a[i, :8])
