#!/usr/bin/env python3
"""Differential fuzz on the GPU: random valid problems x random tuning variants (tests/gpu_util.variants) and the automatic
dispatch, every result compared bit for bit (int32 accumulators and fp16 outputs) with the unsplit stream kernel -- itself
pinned against the CPU oracle by the test-suite.  A second stream keeps the chip loaded with other GEMMs half of the time.
usage: SEED=1 SECONDS=60 python tools/fuzz_families.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import GemmHarness, variants
from qqq_amd import ops, pack as P

dev = torch.device("cuda:0")
seed = int(os.environ.get("SEED", "1")); budget = float(os.environ.get("SECONDS", "60"))
rng = np.random.default_rng(seed)
g = torch.Generator(device="cpu").manual_seed(seed)
side = torch.cuda.Stream(device=dev)
t_end = time.time() + budget
n_prob = n_run = 0
bgA = torch.randint(-128, 128, (96, 2048), generator=g, dtype=torch.int8).to(dev)
bgs1 = (torch.rand((96, 1), generator=g) * 0.05 + 0.001).to(dev)
bg = GemmHarness(P.pack_codes(torch.randint(-7, 8, (2048, 2048), generator=g, dtype=torch.int8).to(dev), False), torch.rand((1, 2048), generator=g) * 1e-4 + 1e-5, None, dev)
while time.time() < t_end:
    grouped = bool(rng.integers(0, 2))
    K = int(rng.integers(1, 65)) * 64 * (2 if grouped else 1)
    if grouped: K = (K // 128) * 128 or 128
    N = int(rng.choice([64, 128, 192, 256, 320, 384, 512, 768, 1024, 1280, 2048, 2304, 4096]))
    if not any(K % tk == 0 and N % tn == 0 for tk, tn in [(64, 256), (128, 128), (128, 64), (64, 128)]):
        continue
    M = int(rng.choice([1, 3, 8, 15, 16, 17, 31, 33, 48, 64, 65, 100, 128, 129, 200, 256, 300, 511, 640, 1000, 1536, 2500]))
    codes = torch.randint(0 if grouped else -8, 16 if grouped else 8, (K, N), generator=g, dtype=torch.int8).to(dev)
    B = P.pack_codes(codes, grouped)
    s2 = torch.rand((1, N), generator=g) * 2e-4 + 1e-5
    s3 = (torch.rand((K // 128, N), generator=g) * 15 + 0.5).half() if grouped else None
    h = GemmHarness(B, s2, s3, dev)
    A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
    s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(dev)
    D0, acc0 = h.run(A, s1, dict(kernel=1, ksplit=1))
    vs = variants(M, K, N)
    picks = [vs[i] for i in rng.choice(len(vs), size=min(10, len(vs)), replace=False)] + [dict()]
    n_prob += 1
    for tune in picks:
        # round 5: uneven K slices -- a random skew (-1 = even, 0 = automatic, 1 ... 12 stages / steps; the host clamps it) on half of the split-capable picks
        if tune and tune.get("kernel") in (1, 4, 5) and "skew" not in tune and rng.random() < 0.5:
            tune = dict(tune, skew=int(rng.integers(-1, 13)))
        loaded = bool(rng.integers(0, 2))
        if loaded:
            with torch.cuda.stream(side):
                for _ in range(4):
                    Db = torch.empty((96, 2048), dtype=torch.float16, device=dev)
                    ops.qqq_gemm_ex(bgA, bg.B, bg.C, Db, bgs1, bg.s2, bg.s3, bg.ws, -1, -1, -1, 16, tune=dict(kernel=4, waves=4, ksplit=2, pf=2))
        try:
            D, acc = h.run(A, s1, tune or None)
        except RuntimeError as e:
            print("ERROR", (M, N, K), grouped, tune, str(e)[:120]); continue
        n_run += 1
        if not (np.array_equal(acc, acc0) and np.array_equal(D.view(np.uint16), D0.view(np.uint16))):
            bad = np.argwhere(acc != acc0)
            print("MISMATCH", (M, N, K), "g128" if grouped else "pc", tune, "loaded" if loaded else "idle", "n bad", len(bad), bad[:3].tolist(), flush=True)
    assert int(h.ws.abs().max()) == 0, ("workspace not zero", (M, N, K), grouped)
    torch.cuda.synchronize()
print(f"fuzz: {n_prob} problems, {n_run} runs, seed {seed}: done")
