#!/usr/bin/env python3
"""Differential fuzz of the wide kernel's persistent tile walk on the GPU: random LARGE problems -- between one
and four tiles per workgroup of the walk's grid, any K from 8 to 48 stages (so that seams fall behind every stage position and
the slow path of the running offsets is entered at every phase), ragged last m-tiles, last strips that overhang n, with and
without a bias -- in the three tile shapes, forced (glds=2) and as the automatic dispatch picks, every result compared bit for
bit (int32 accumulators and fp16 outputs) with the tiled kernel, which the parity tests pin against the CPU oracle.  Half of the
launches run while a second stream keeps the chip loaded (workgroups of the walk then share their CUs with other kernels).
usage: SEED=1 SECONDS=60 python tools/fuzz_walk.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import GemmHarness
from qqq_amd import _lib, ops, pack as P

dev = torch.device("cuda:0")
seed = int(os.environ.get("SEED", "1")); budget = float(os.environ.get("SECONDS", "60"))
rng = np.random.default_rng(seed)
g = torch.Generator(device="cpu").manual_seed(seed)
side = torch.cuda.Stream(device=dev)
cus = torch.cuda.get_device_properties(dev).multi_processor_count & ~7
t_end = time.time() + budget
n_prob = n_run = n_walk = 0
bgA = torch.randint(-128, 128, (96, 2048), generator=g, dtype=torch.int8).to(dev)
bgs1 = (torch.rand((96, 1), generator=g) * 0.05 + 0.001).to(dev)
bg = GemmHarness(P.pack_codes(torch.randint(-7, 8, (2048, 2048), generator=g, dtype=torch.int8).to(dev), False), torch.rand((1, 2048), generator=g) * 1e-4 + 1e-5, None, dev)
while time.time() < t_end:
    grouped = bool(rng.integers(0, 2))
    K = 128 * int(rng.integers(8, 49))
    N = 64 * int(rng.integers(32, 97))                        # 2048 ... 6144, any multiple of 64: strips overhang
    strips = -(-N // 256)
    per_wg = float(rng.uniform(1.0, 4.0))
    M = 256 * max(1, int(per_wg * cus / strips)) + int(rng.integers(0, 256))
    if M * K > 200e6 or M * N > 80e6:
        continue
    codes = torch.randint(0 if grouped else -8, 16 if grouped else 8, (K, N), generator=g, dtype=torch.int8).to(dev)
    B = P.pack_codes(codes, grouped)
    s2 = torch.rand((1, N), generator=g) * 2e-4 + 1e-5
    s3 = (torch.rand((K // 128, N), generator=g) * 15 + 0.5).half() if grouped else None
    h = GemmHarness(B, s2, s3, dev)
    A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
    s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(dev)
    bias = (torch.randn(N, generator=g) * 0.1).half().numpy() if rng.integers(0, 2) else None
    ref = dict(kernel=2, bm=256, glds=1, stages=5, ksplit=1)
    D0, acc0 = h.run(A, s1, ref)
    D0b = h.run(A, s1, ref, want_acc=False, bias=bias)[0] if bias is not None else None
    n_prob += 1
    for tune in (dict(kernel=5, glds=2), dict(kernel=5, glds=2, mt=8), dict(kernel=5, glds=2, bm=128), dict(kernel=5), dict()):
        pl = _lib.plan(M, N, K, 128 if grouped else -1, 16, tune=tune or None)
        if tune.get("glds") == 2:
            tiles = -(-M // (16 * (tune.get("mt") or 16))) * -(-N // (tune.get("bm") or 256))
            if tiles >= cus and not (pl["kernel"] == 5 and pl["glds"] == 2):
                print("ERROR plan is not the tile walk", (M, N, K), tune, pl); continue
        n_walk += int(pl["kernel"] == 5 and pl["glds"] == 2)
        loaded = bool(rng.integers(0, 2))
        if loaded:
            with torch.cuda.stream(side):
                for _ in range(6):
                    Db = torch.empty((96, 2048), dtype=torch.float16, device=dev)
                    ops.qqq_gemm_ex(bgA, bg.B, bg.C, Db, bgs1, bg.s2, bg.s3, bg.ws, -1, -1, -1, 16, tune=dict(kernel=4, waves=4, ksplit=2, pf=2))
        try:
            D, acc = h.run(A, s1, tune or None)
            Db_ = h.run(A, s1, tune or None, want_acc=False, bias=bias)[0] if bias is not None else None
        except RuntimeError as e:
            print("ERROR", (M, N, K), grouped, tune, str(e)[:120]); continue
        n_run += 1
        ok = np.array_equal(acc, acc0) and np.array_equal(D.view(np.uint16), D0.view(np.uint16))
        if bias is not None:
            ok = ok and np.array_equal(Db_.view(np.uint16), D0b.view(np.uint16))
        if not ok:
            bad = np.argwhere(acc != acc0)
            print("MISMATCH", (M, N, K), "g128" if grouped else "pc", tune, pl, "loaded" if loaded else "idle", "n bad", len(bad), bad[:3].tolist(), flush=True)
    assert int(h.ws.abs().max()) == 0, ("workspace not zero", (M, N, K), grouped)
    torch.cuda.synchronize()
    del h, A, B, codes
print(f"fuzz_walk: {n_prob} problems, {n_run} runs, {n_walk} of them the tile walk, seed {seed}: done")
