#!/usr/bin/env python3
"""Offline check of the dispatcher against `tools/dispatch_check.py` outputs (profiles/r04_dispatch_check_*.txt): for every measured
line, which family the CURRENT library plans (qqq_w4a8_plan: pure host logic, no GPU needed) and how far that family's measured
time is from the best measured one.  The measurements stay what they were on their box; only the plan is re-evaluated -- so a
change to a cost model can be judged (and regression-tested, tests/test_dispatch_regret_cpu.py) without a visit.

    python tools/dispatch_regret.py profiles/r04_dispatch_check_final.txt [more files]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LINE = re.compile(r"N=\s*(\d+) K=\s*(\d+) (\w+)\s+M=\s*(\d+)\s+auto\(k(\d),ks(\d+).*?\)\s+([\d.]+) \|(.*)")
FAMILY = {1: "stream", 2: "tiled", 3: "column", 4: "panel", 5: "wide"}
WIDE = {(16, 256, 1): "wide", (16, 256, 2): "w16x2", (8, 256, 1): "w8", (16, 128, 1): "w128", (16, 128, 2): "w128x2"}


def grid_files(root=ROOT):
    """the committed dispatch grids, ONE file per grid name: the newest round's measurement of it (profiles/r06_dispatch_check_<name>.txt where round 6 measured the
    grid again -- the seven grids that price the wide kernel, whose loop changed -- otherwise round 5's)"""
    import glob

    best = {}
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r0[5-9]_dispatch_check_*.txt"))):
        m = re.match(r"r(\d+)_dispatch_check_(.*)\.txt", os.path.basename(f))
        if m and (m.group(2) not in best or int(m.group(1)) > best[m.group(2)][0]):
            best[m.group(2)] = (int(m.group(1)), f)
    return [best[k][1] for k in sorted(best)]


def column_of(plan, measured, M=0):
    """the dispatch_check column a plan corresponds to"""
    name = FAMILY[plan["kernel"]]
    if name == "panel" and plan["bm"] == 256:
        name = "panel256x2" if plan["pw"] == 2 else "panel256"
    if name == "panel" and plan["mt"] == 4 and M > 64:
        name = "panel64"  # several 64-token m-blocks (round 5): a column of its own -- the round-4 files measured "panel" with 128-token m-blocks
    if name == "wide":
        name = WIDE.get((plan["mt"], plan["bm"], plan["ksplit"]), "wide")
        if plan["glds"] == 2 and "walk" in measured:
            name = "walk"
    return name


def regrets(path):
    """[(N, K, mode, M, planned column, its measured us, best measured us)] for every line of a dispatch_check output whose planned
    variant was measured"""
    from qqq_amd import _lib

    out = []
    for line in open(path):
        m = LINE.match(line)
        if not m:
            continue
        N, K, mode, M = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
        cells = m.group(8).split("<--")[0].split()
        measured = {cells[i]: float(cells[i + 1]) for i in range(0, len(cells) - 1, 2)}
        measured = {k: v for k, v in measured.items() if v == v}  # (nan: variant not applicable at this point)
        plan = _lib.plan(M, N, K, 128 if mode == "g128" else -1, 16)
        col = column_of(plan, measured, M)
        if col in measured:
            out.append((N, K, mode, M, col, measured[col], min(measured.values())))
    return out


def summary(rows, flag=0.03):
    """(points, points above `flag`, worst regret)"""
    rel = [got / best - 1.0 for *_, got, best in rows]
    return len(rel), sum(r > flag for r in rel), max(rel, default=0.0)


if __name__ == "__main__":
    rows = [r for f in sys.argv[1:] for r in regrets(f)]
    for N, K, mode, M, col, got, best in rows:
        if got / best - 1.0 > 0.03:
            print(f"N={N:5d} K={K:5d} {mode:4s} M={M:5d}  planned {col:10s} {got:8.1f} us   best measured {best:8.1f}   regret {100 * (got / best - 1):.0f} %")
    n, above, worst = summary(rows)
    print(f"{n} points, {above} above 3 %, worst {100 * worst:.1f} %")
