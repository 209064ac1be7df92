#!/usr/bin/env python3
"""Is the automatic dispatch picking the fastest family?  For each layer shape / mode / m: auto vs every family forced
(stream, tiled, panel with its own defaults).  Output kept under profiles/."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
from qqq_amd import _lib

dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "8192x21760,4096x4096,11008x4096,4096x11008").split(",")]
Ms = [int(x) for x in os.environ.get("MS", "32,48,64,96,128,160,256,320,384,512").split(",")]
modes = os.environ.get("MODES", "pc,g128").split(",")
iters = int(os.environ.get("ITERS", "10"))
for (N, K) in shapes:
    for mode in modes:
        grouped = mode.startswith("g128")
        expand = mode == "g128x"  # per-group with the opt-in expanded int8 weights (every weight copy expanded; every variant's calls get them)
        # enough weight copies that a rotation is longer (1.1 GB) than what the 256 MB Infinity Cache can hold: below ~64 tokens a launch is one pass over the weights,
        # and with the 4 / 12 copies of rounds 2-4 the plain-load kernels were partly served from that cache (profiles/r05_nt_weight_loads.txt)
        layer = Bn.Layer(dev, grouped=grouped, nbuf=Bn.copies_for(N, K) if min(Ms) <= 128 else (4 if N * K > 6e7 else 12), N=N, K=K, expand=expand)
        for M in Ms:
            A, s1 = Bn.make_tokens(dev, M, M, K=K)
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            # every variant timed in turn, `ROUNDS` times over (the first variant of a point otherwise pays for the cold A / D /
            # page tables of that point: "auto" read 5-9 % slower than the identical forced launch in the first version)
            cands = {"auto": None, "column": dict(kernel=3) if M <= 32 else 0, "stream": dict(kernel=1) if M <= 256 else 0,
                     "tiled": dict(kernel=2) if (M > 32 and K % 128 == 0 and os.environ.get("TILED") == "1") else 0,  # (no longer a candidate of the automatic dispatch)
                     "panel": dict(kernel=4, mt=8 if M > 64 else 0) if M > 8 else 0,
                     "panel256": dict(kernel=4, bm=256) if M > 8 else 0, "panel64": dict(kernel=4, mt=4) if 64 < M <= 1024 else 0, "panel256x2": dict(kernel=4, bm=256, pw=2, pf=4) if M >= 256 else 0,
                     "wide": dict(kernel=5) if M > 256 else 0}
            if os.environ.get("WIDE_SHAPES") == "1" and M > 256:  # every shape of the wide kernel, for fitting its cost model
                cands.update({"w16x2": dict(kernel=5, ksplit=2), "w8": dict(kernel=5, mt=8), "w128": dict(kernel=5, bm=128),
                              "w128x2": dict(kernel=5, bm=128, ksplit=2), "walk": dict(kernel=5, glds=2), "plain": dict(kernel=5, glds=1)})
            cands = {k: v for k, v in cands.items() if v != 0}
            samples = {k: [] for k in cands}
            for k, tune in cands.items():
                layer.time_calls(A, s1, D, 2, tune=tune)
            # (the weight copies rotate on from group to group -- bench.Layer.time_calls -- and there are enough of them, see nbuf above)
            # ... in a different (seeded) order every round: on a power-limited chip a variant timed right behind a slow, cool one finds higher clocks -- with a fixed order
            # the same plan read 502 us as column `wide` (behind the panel variants) and 536 / 538 as `walk` / auto at 8192 tokens (profiles/r05_dispatch_check_merged.txt of visit 27)
            import random
            for r in range(int(os.environ.get("ROUNDS", "3"))):
                order = list(cands.items())
                random.Random(hash((N, K, M, grouped, r)) & 0xffffffff).shuffle(order)
                for k, tune in order:
                    samples[k].extend(layer.time_calls(A, s1, D, max(2, iters // 3), tune=tune) * 1e3)
            p = _lib.plan(M, N, K, 128 if grouped else -1, 16, tune=dict(w8=1) if expand else None)
            res = {k: float(np.median(v)) for k, v in samples.items()}
            for k in ("column", "stream", "tiled", "panel", "panel256", "panel256x2", "panel64", "wide"):
                res.setdefault(k, float("nan"))
            best = min((v, k) for k, v in res.items() if v == v and k != "auto")
            flag = "" if res["auto"] <= best[0] * 1.03 else f"   <-- {best[1]} is {100 * (res['auto'] / best[0] - 1):.0f}% faster"
            extra = "".join(f" {k} {res[k]:7.1f}" for k in ("w16x2", "w8", "w128", "w128x2", "walk", "plain") if k in res)
            print(f"N={N:5d} K={K:5d} {mode:5s} M={M:4d}  auto(k{p['kernel']},ks{p['ksplit']}{',walk' if p['kernel'] == 5 and p['glds'] == 2 else ''}{',split@' + str(p['split_m']) if p.get('split_m') else ''}) {res['auto']:7.1f} | column {res['column']:7.1f} stream {res['stream']:7.1f} tiled {res['tiled']:7.1f} panel {res['panel']:7.1f} panel256 {res['panel256']:7.1f} panel256x2 {res['panel256x2']:7.1f} panel64 {res['panel64']:7.1f} wide {res['wide']:7.1f}{extra}{flag}")
            sys.stdout.flush()
        del layer
        torch.cuda.empty_cache()
