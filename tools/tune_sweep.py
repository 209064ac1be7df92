#!/usr/bin/env python3
"""Times every kernel variant at the BASELINE sizes (cold: rotating weight buffers) and prints a table.
Used to pick the host dispatch heuristics in qqq_w4a8.hip; output is kept under profiles/."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn  # noqa: E402


def variants(M, grouped, N=8192):
    v = []
    if M <= 128:
        for waves in (4, 8, 16):
            if waves == 16 and M > 16:
                continue
            for ks in (1, 2, 3, 4, 5, 6, 8, 16):
                if ks * M > 1024:
                    continue
                for pf in ((3, 5, 7) if M <= 16 else (3, 5) if M <= 32 else (0,)):
                    if waves == 16 and pf != 3:
                        continue
                    for fused in ((1, 2) if ks > 1 else (1,)):
                        v.append(dict(kernel=1, waves=waves, ksplit=ks, fused=fused, pf=pf))
    if M <= 32 and N % 64 == 0:
        for pf in (2, 3, 4, 6, 8):
            v.append(dict(kernel=3, pf=pf, ksplit=1))
    if M >= 16:
        for bm in (64, 128, 131, 130, 256, 258, 259):
            rows = 256 if bm >= 256 else (128 if bm >= 128 else 64)
            if rows > 64 and M < rows // 2:
                continue
            for stages in (0, 2, 3, 4, 5, 6):
                if stages == 4 and rows == 256:
                    continue
                if stages == 5 and bm in (64, 128):
                    continue
                if stages == 6 and bm != 256:
                    continue
                for ks in (1, 2, 3, 4, 8):
                    tiles = -(-M // rows) * -(-N // 256)
                    if ks > 1 and (tiles * rows * 256 > 1024 * N or tiles >= 256 or tiles * ks > 512):
                        continue  # in-launch split-K needs one tile-sized slot per tile in the 1024 rows of C
                    v.append(dict(kernel=2, bm=bm, glds=(2 if stages == 0 else 1), stages=stages, ksplit=ks))
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--ms", type=str, default="1,16,64,128,256,1024,4096")
    ap.add_argument("--modes", type=str, default="pc,g128")
    ap.add_argument("--nk", type=str, default=f"{Bn.N_FULL},{Bn.K_FULL}", help="layer shape N,K")
    ap.add_argument("--top", type=int, default=12)
    args = ap.parse_args()
    NN, KK = [int(x) for x in args.nk.split(",")]
    dev = torch.device("cuda:0")
    Ms = [int(x) for x in args.ms.split(",")]
    for mode in args.modes.split(","):
        grouped = mode == "g128"
        layer = Bn.Layer(dev, grouped=grouped, N=NN, K=KK, nbuf=max(4, min(40, (400 << 20) // (NN * KK // 2))))
        for M in Ms:
            A, s1 = Bn.make_tokens(dev, M, M, K=KK)
            D = torch.empty((M, NN), dtype=torch.float16, device=dev)
            rows = []
            for tune in [None] + variants(M, grouped, NN):
                try:
                    layer.time_calls(A, s1, D, 2, tune=tune)
                    t = layer.time_calls(A, s1, D, args.iters, tune=tune) * 1e3
                    rows.append((float(np.mean(t)), float(np.min(t)), tune))
                except Exception as e:
                    rows.append((float("inf"), float("inf"), dict(error=str(e), tune=tune)))
            ops = Bn.algorithmic_ops(M, NN, KK)
            byts = Bn.algorithmic_bytes(M, NN, KK, grouped)
            auto = rows[0]
            print(f"== N={NN} K={KK} mode={mode} M={M}: auto {auto[0]:.1f} us  ({ops/auto[0]/1e6:.1f} TOPS, {byts/auto[0]/1e3:.0f} GB/s)")
            for mean, mn, tune in sorted(rows[1:], key=lambda r: r[0])[:args.top]:
                print(f"   {mean:9.1f} us (min {mn:8.1f})  {ops/mean/1e6:8.1f} TOPS {byts/mean/1e3:7.0f} GB/s  {tune}")
            sys.stdout.flush()
        del layer
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
