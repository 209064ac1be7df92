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
        grouped = mode == "g128"
        layer = Bn.Layer(dev, grouped=grouped, nbuf=4 if N * K > 6e7 else 12, N=N, K=K)
        for M in Ms:
            A, s1 = Bn.make_tokens(dev, M, M, K=K)
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            def t(tune):
                layer.time_calls(A, s1, D, 3, tune=tune)
                return float(np.median(layer.time_calls(A, s1, D, iters, tune=tune) * 1e3))
            p = _lib.plan(M, N, K, 128 if grouped else -1, 16)
            res = {"auto": t(None), "column": t(dict(kernel=3)) if M <= 32 else float("nan"), "stream": t(dict(kernel=1)) if M <= 256 else float("nan"), "tiled": t(dict(kernel=2)) if (M > 32 and K % 128 == 0) else float("nan"),
                   "panel": t(dict(kernel=4)) if M > 8 else float("nan"), "panel256": t(dict(kernel=4, bm=256)) if M > 8 else float("nan"),
                   "panel256x2": t(dict(kernel=4, bm=256, pw=2, pf=4)) if M >= 256 else float("nan"),
                   "wide": t(dict(kernel=5)) if M > 256 else float("nan")}
            if os.environ.get("WIDE_SHAPES") == "1" and M > 256:  # every shape of the wide kernel, for fitting its cost model
                res.update({"w16x2": t(dict(kernel=5, ksplit=2)), "w8": t(dict(kernel=5, mt=8)), "w128": t(dict(kernel=5, bm=128)),
                            "w128x2": t(dict(kernel=5, bm=128, ksplit=2))})
            best = min((v, k) for k, v in res.items() if v == v and k != "auto")
            flag = "" if res["auto"] <= best[0] * 1.03 else f"   <-- {best[1]} is {100 * (res['auto'] / best[0] - 1):.0f}% faster"
            extra = "".join(f" {k} {res[k]:7.1f}" for k in ("w16x2", "w8", "w128", "w128x2") if k in res)
            print(f"N={N:5d} K={K:5d} {mode:4s} M={M:4d}  auto(k{p['kernel']},ks{p['ksplit']}) {res['auto']:7.1f} | column {res['column']:7.1f} stream {res['stream']:7.1f} tiled {res['tiled']:7.1f} panel {res['panel']:7.1f} panel256 {res['panel256']:7.1f} panel256x2 {res['panel256x2']:7.1f} wide {res['wide']:7.1f}{extra}{flag}")
            sys.stdout.flush()
        del layer
        torch.cuda.empty_cache()
