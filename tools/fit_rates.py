#!/usr/bin/env python3
"""Generates qqq_amd/csrc/qqq_rates.h -- the cost tables of the panel and the wide kernel -- from the committed dispatch checks (CPU only).

Every forced panel variant measured in profiles/r05_dispatch_check_*.txt (columns panel / panel256 / panel256x2: 128-column strips, 256-column strips,
256-column strips with 64 columns per wave) is grouped by (shape, 16-token tiles per m-block, mode) and fitted, per group, with ONE linear form

    us = rounds * (a + c * [ksplit > 1] + d * max(0, ksplit - 2) + b * stages_per_workgroup),    rounds = ceil(workgroups / 256)

by least squares on the relative error (a: launch, fill, epilogue; c, d: the in-launch split-K hand-off; b: microseconds per 128-k stage).  The K split of
each measured point is the one the library plans for that forced variant (`qqq_w4a8_plan`; it does not depend on this table).  A kernel change that moves
these rates is followed by: re-measure (tools/visits/r5_v07.sh), re-run this tool, rebuild -- no constant is edited by hand.

The wide kernel's three tile shapes (256 x 256, 256 x 128, 128 x 256; columns wide / w16x2 / w128 / w128x2 / w8 of the checks run with WIDE_SHAPES=1) get the
three rates of wide_estimate's form -- us = 3.7 + rounds * (fixed + handoff * [ksplit > 1] * tile KiB / 256 + stages per workgroup * t_stage * load) with the
rounds / load rules of qqq_w4a8.hip -- fitted the same way per (shape, mode).

    python tools/fit_rates.py            # rewrites the header, prints the fit quality per group"""
import glob
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "qqq_amd", "csrc", "qqq_rates.h")
COLS = (("panel", lambda M: dict(kernel=4, mt=8 if M > 64 else 0)), ("panel256", lambda M: dict(kernel=4, bm=256)),
        ("panel256x2", lambda M: dict(kernel=4, bm=256, pw=2, pf=4)))
MTS = (1, 2, 4, 8)


def collect(files):
    import dispatch_regret as R
    from qqq_amd import _lib

    data = {}
    for f in files:
        for line in open(f):
            m = R.LINE.match(line)
            if not m:
                continue
            N, K, mode, M = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
            cells = m.group(8).split("<--")[0].split()
            meas = {cells[i]: float(cells[i + 1]) for i in range(0, len(cells) - 1, 2)}
            for ci, (col, tune) in enumerate(COLS):
                v = meas.get(col)
                if v is None or v != v:
                    continue
                p = _lib.plan(M, N, K, 128 if mode == "g128" else -1, 16, tune=tune(M))
                rows, bn, ks = 16 * p["mt"], p["bm"], p["ksplit"]
                wgs = ((M + rows - 1) // rows) * ((N + bn - 1) // bn) * ks
                rounds = math.ceil(wgs / 256)
                nst = (K // 64 + 1) // 2
                data.setdefault((ci, p["mt"], mode == "g128"), []).append(([rounds, rounds * (ks > 1), rounds * max(0, ks - 2), rounds * nst / ks], v))
    return data


WIDE_COLS = {"wide": dict(kernel=5, glds=1), "w16x2": dict(kernel=5, ksplit=2, glds=1), "w8": dict(kernel=5, mt=8, glds=1),
             "w128": dict(kernel=5, bm=128, glds=1), "w128x2": dict(kernel=5, bm=128, ksplit=2, glds=1)}
WIDE_LAUNCH, WIDE_LO = 3.7, (0.80, 0.85)  # the part of wide_estimate's form that is not fitted: launch, and the stage-time factor of a half-empty chip (per mode)


def wide_features(tl, ks, rows, bn, nst, grouped):
    """(rounds, rounds * [split] * tile KiB / 256, rounds * stages per workgroup * load) -- wide_estimate's rounds / load rules (qqq_w4a8.hip)"""
    x = tl * ks / 256.0
    cx = math.ceil(x)
    rounds = 1.0 if x <= 1.0 else cx - 0.3 * (cx - x)
    lo = WIDE_LO[1 if grouped else 0]
    rel = 0.0 if x <= 0.5 else (x - 0.5) / 0.5
    load = lo + (1.0 - lo) * rel * rel if x <= 1.0 else 1.0
    return [rounds, rounds * (ks > 1) * rows * bn / 65536.0, rounds * nst / ks * load]


def collect_wide(files):
    import dispatch_regret as R
    from qqq_amd import _lib

    data = {}
    for f in files:
        for line in open(f):
            m = R.LINE.match(line)
            if not m or "w16x2" not in line:
                continue
            N, K, mode, M = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
            if K % 128:
                continue
            cells = m.group(8).split("<--")[0].split()
            meas = {cells[i]: float(cells[i + 1]) for i in range(0, len(cells) - 1, 2)}
            for col, tune in WIDE_COLS.items():
                v = meas.get(col)
                if v is None or v != v:
                    continue
                p = _lib.plan(M, N, K, 128 if mode == "g128" else -1, 16, tune=tune)
                if p["kernel"] != 5 or p["glds"] == 2 or (col.endswith("x2") and p["ksplit"] != 2):
                    continue
                rows, bn, ks = 16 * p["mt"], p["bm"], p["ksplit"]
                tl = ((M + rows - 1) // rows) * ((N + bn - 1) // bn)
                shape = 0 if (p["mt"] == 16 and bn == 256) else 1 if p["mt"] == 16 else 2
                data.setdefault((shape, mode == "g128"), []).append((wide_features(tl, ks, rows, bn, K // 128, mode == "g128"), v))
    return data


def fit_wide(rows):
    A = np.array([r[0] for r in rows], float)
    v = np.array([r[1] for r in rows], float)
    w = 1.0 / v
    if not A[:, 1].any():  # a shape that was never measured split: its hand-off rate is the 256 x 128 tiles' (set by the caller)
        c3 = np.linalg.lstsq((A[:, [0, 2]]) * w[:, None], (v - WIDE_LAUNCH) * w, rcond=None)[0]
        coef = np.array([c3[0], 0.0, c3[1]])
    else:
        coef = np.linalg.lstsq(A * w[:, None], (v - WIDE_LAUNCH) * w, rcond=None)[0]
    e = (A @ coef + WIDE_LAUNCH) / v - 1.0
    return coef, len(v), float(np.abs(e).mean()), float(e.mean()), float(np.abs(e).max())


def fit(rows):
    A = np.array([r[0] for r in rows], float)
    y = np.array([r[1] for r in rows], float)
    w = 1.0 / y
    coef = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
    if not A[:, 1].any() or (A[:, 1] == A[:, 0]).all():  # the group was only ever measured split (or only unsplit): a and c are not separable -- share the sum
        s = coef[0] + coef[1]
        coef[0], coef[1] = 0.75 * s, 0.25 * s
    e = (A @ coef) / y - 1.0
    return coef, len(y), float(np.abs(e).mean()), float(e.mean()), float(np.abs(e).max())


def main():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_dispatch_check_*.txt")))
    data = collect(files)
    names = ("128-column strips, 32 columns per wave", "256-column strips, 32 columns per wave", "256-column strips, 64 columns per wave (128-token m-blocks only)")
    lines = ["// qqq_rates.h -- GENERATED by tools/fit_rates.py from profiles/r05_dispatch_check_*.txt (%d files); do not edit by hand." % len(files),
             "// panel kernel: us = rounds * (a + c * [ksplit > 1] + d * max(0, ksplit - 2) + b * stages per workgroup), rounds = ceil(workgroups / 256); per group: points, mean |error|, worst.",
             "#ifndef QQQ_AMD_QQQ_RATES_H_", "#define QQQ_AMD_QQQ_RATES_H_", "",
             "struct QqqPanelRate { double a, c, d, b; };", "// [shape][log2(16-token tiles per m-block)][per-group]", "static const QqqPanelRate kQqqPanelRates[3][4][2] = {"]
    for ci in range(3):
        lines.append("    {  // " + names[ci])
        for mt in MTS:
            cells = []
            for g in (False, True):
                rows = data.get((ci, mt, g))
                if rows and len(rows) >= 8:
                    coef, n, mae, bias, worst = fit(rows)
                    cells.append("{%.3f, %.3f, %.3f, %.4f}" % tuple(coef))
                    print(f"{COLS[ci][0]:11s} mt={mt} {'g128' if g else 'pc  '} n={n:4d}  a={coef[0]:6.2f} c={coef[1]:5.2f} d={coef[2]:5.2f} b={coef[3]:.4f}   mean |err| {100 * mae:4.1f}%  bias {100 * bias:+4.1f}%  worst {100 * worst:4.1f}%")
                    note = f"{n} points, {100 * mae:.1f} %, {100 * worst:.0f} %"
                else:
                    cells.append("{0.0, 0.0, 0.0, 0.0}")
                    note = "not measured: never planned"
                cells[-1] += "  /* %s: %s */" % ("per-group" if g else "per-channel", note)
            lines.append("        {%s,\n         %s},  // %d tokens per m-block" % (cells[0], cells[1], 16 * mt))
        lines.append("    },")
    lines += ["};", ""]
    wide = collect_wide(files)
    wnames = ("256 x 256 tiles", "256 x 128 tiles (32 columns per wave)", "128 x 256 tiles")
    lines += ["// wide kernel: us = 3.7 + rounds * (fixed + handoff * [ksplit > 1] * tile KiB / 256 + stages per workgroup * t_stage * load); rounds / load: wide_estimate (qqq_w4a8.hip)",
              "struct QqqWideRate { double fixed, handoff, t_stage; };", "// [shape][per-group]", "static const QqqWideRate kQqqWideRates[3][2] = {"]
    fitted = {}
    for shape in range(3):
        for g in (False, True):
            fitted[(shape, g)] = fit_wide(wide[(shape, g)])
    for shape in range(3):
        cells = []
        for g in (False, True):
            coef, n, mae, bias, worst = fitted[(shape, g)]
            if coef[1] == 0.0:
                coef = np.array([coef[0], fitted[(1, g)][0][1], coef[2]])  # (never measured split)
            cells.append("{%.3f, %.3f, %.4f}  /* %s: %d points, %.1f %%, %.0f %% */" % (coef[0], coef[1], coef[2], "per-group" if g else "per-channel", n, 100 * mae, 100 * worst))
            print(f"wide shape {shape} {'g128' if g else 'pc  '} n={n:4d}  fixed={coef[0]:6.2f} handoff={coef[1]:5.2f} t_stage={coef[2]:.4f}   mean |err| {100 * mae:4.1f}%  bias {100 * bias:+4.1f}%  worst {100 * worst:4.1f}%")
        lines.append("    {%s,\n     %s},  // %s" % (cells[0], cells[1], wnames[shape]))
    lines += ["};", "", "#endif  // QQQ_AMD_QQQ_RATES_H_", ""]
    open(OUT, "w").write("\n".join(lines))
    print("wrote", os.path.relpath(OUT, ROOT))


if __name__ == "__main__":
    main()
