#!/usr/bin/env python3
"""ONE tool that holds every cost model of the dispatcher against every committed hardware measurement (CPU only).

For each line of the `tools/dispatch_check.py` outputs under profiles/ (automatic choice + every family forced, timed interleaved on an MI355X) the CURRENT
library's models are asked for their price of each family (`qqq_w4a8_model_us`, pure host logic) and compared with the fastest measured variant of that
family at that point.  Per family and token regime: number of points, mean |error|, bias (model / measured - 1, mean) and the worst point.

    python tools/cost_model_report.py [files ...]  > profiles/r05_cost_model_error.txt        (default: every profiles/r05_dispatch_check_*.txt)

What the numbers are for: a model only has to ORDER the families correctly (tests/test_dispatch_regret_cpu.py checks that on the same files); this report
says how far each is from the clock -- which constants a kernel change has moved, and by how much -- without a refit by eye."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FAMILY_OF = {"column": "column", "stream": "stream", "panel": "panel", "panel256": "panel", "panel256x2": "panel", "panel64": "panel",
             "wide": "wide", "w16x2": "wide", "w8": "wide", "w128": "wide", "w128x2": "wide", "walk": "wide", "plain": "wide"}
REGIMES = ((1, 8, "1-8"), (9, 32, "9-32"), (33, 64, "33-64"), (65, 256, "65-256"), (257, 1024, "257-1024"), (1025, 1 << 30, ">1024"))


def points(path):
    """[(N, K, mode, M, family, model us, fastest measured us of that family)]"""
    import dispatch_regret as R
    from qqq_amd import _lib

    out = []
    for line in open(path):
        m = R.LINE.match(line)
        if not m:
            continue
        N, K, mode, M = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
        cells = m.group(8).split("<--")[0].split()
        measured = {cells[i]: float(cells[i + 1]) for i in range(0, len(cells) - 1, 2)}
        best = {}
        for col, us in measured.items():
            fam = FAMILY_OF.get(col)
            if fam and us == us and not (fam == "wide" and M <= 1024 and "w128" not in measured):  # (a file without the wide shapes: the lone 256 x 256 column is no measure of the family)
                best[fam] = min(best.get(fam, 1e30), us)
        model = _lib.model_us(M, N, K, 128 if mode == "g128" else -1, 16)
        for fam, us in model.items():
            if fam in best:
                out.append((N, K, mode, M, fam, us, best[fam]))
    return out


def table(rows):
    """{(family, regime): (n, mean |err|, bias, worst err, worst point)}"""
    acc = {}
    for N, K, mode, M, fam, mod, meas in rows:
        reg = next(name for lo, hi, name in REGIMES if lo <= M <= hi)
        acc.setdefault((fam, reg), []).append((mod / meas - 1.0, (N, K, mode, M, mod, meas)))
    res = {}
    for key, v in acc.items():
        errs = [e for e, _ in v]
        worst = max(v, key=lambda x: abs(x[0]))
        res[key] = (len(v), sum(abs(e) for e in errs) / len(errs), sum(errs) / len(errs), worst[0], worst[1])
    return res


def main(files):
    rows = [r for f in files for r in points(f)]
    print(f"# {len(rows)} (point, family) pairs from {len(files)} files; model = qqq_w4a8_model_us of the current library, measured = fastest forced variant of the family")
    print(f"# {'family':8s} {'tokens':9s} {'n':>5s} {'mean|err|':>10s} {'bias':>8s} {'worst':>8s}   worst point (N, K, mode, M: model vs measured us)")
    t = table(rows)
    for fam in ("column", "stream", "panel", "wide"):
        for _, _, reg in REGIMES:
            if (fam, reg) in t:
                n, mae, bias, w, (N, K, mode, M, mod, meas) = t[(fam, reg)]
                print(f"  {fam:8s} {reg:9s} {n:5d} {100 * mae:9.1f}% {100 * bias:+7.1f}% {100 * w:+7.1f}%   N={N} K={K} {mode} M={M}: {mod:.1f} vs {meas:.1f}")
    return t


if __name__ == "__main__":
    import dispatch_regret as R

    files = sys.argv[1:] or R.grid_files()
    main(files)
