#!/usr/bin/env python3
"""Panel-kernel tuning sweep at the BASELINE layer (N=8192, K=21760): every shape x split for m = 16..256, against the
automatic dispatch.  Output is kept under profiles/."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn

dev = torch.device("cuda:0")
modes = os.environ.get("MODES", "pc,g128").split(",")
Ms = [int(x) for x in os.environ.get("MS", "16,32,64,128").split(",")]
iters = int(os.environ.get("ITERS", "12"))
for mode in modes:
    grouped = mode == "g128"
    layer = Bn.Layer(dev, grouped=grouped, nbuf=4)
    for M in Ms:
        A, s1 = Bn.make_tokens(dev, M, M)
        D = torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev)
        def t(tune):
            layer.time_calls(A, s1, D, 3, tune=tune)
            v = layer.time_calls(A, s1, D, iters, tune=tune) * 1e3
            return float(np.median(v)), float(v.min())
        rows = []
        rows.append(("auto", t(None)))
        mts = [m for m in (1, 2, 4, 8) if 16 * m >= min(M, 128)][:1]
        for mt in mts:
            for bm, waves in ((128, 8), (256, 8), (128, 4)):
                for ks in ((1, 2, 3, 4, 5, 6, 8) if bm == 128 else (2, 4, 6, 8, 10, 12, 16)):
                    for pf in ((2, 4, 8) if mt <= 4 else (2, 3, 4)):
                        tune = dict(kernel=4, mt=mt, bm=bm, waves=waves, ksplit=ks, pf=pf)
                        try:
                            rows.append((str(tune), t(tune)))
                        except Exception as e:
                            rows.append((str(tune) + " ERR " + str(e)[:60], (1e9, 1e9)))
        rows.sort(key=lambda r: r[1][0])
        ops = Bn.algorithmic_ops(M, Bn.N_FULL, Bn.K_FULL); byts = Bn.algorithmic_bytes(M, Bn.N_FULL, Bn.K_FULL, grouped)
        print(f"== mode={mode} M={M}")
        for name, (med, mn) in rows[:14]:
            print(f"   {med:8.1f} us (min {mn:8.1f})  {ops/med/1e6:8.1f} TOPS {byts/med/1e3:7.0f} GB/s  {name}")
        auto = [r for r in rows if r[0] == "auto"][0]
        print(f"   auto: {auto[1][0]:.1f} us")
        sys.stdout.flush()
    del layer
    torch.cuda.empty_cache()
