#!/usr/bin/env python3
"""Generates qqq_amd/csrc/qqq_rates.h -- the cost tables of the panel and the wide kernel and the small-m forms of the column and stream kernels -- from the committed
dispatch checks (CPU only).

Every forced panel variant measured in profiles/r05_dispatch_check_*.txt (columns panel / panel256 / panel256x2: 128-column strips, 256-column strips,
256-column strips with 64 columns per wave) is grouped by (shape, 16-token tiles per m-block, mode) and fitted, per group, with ONE linear form

    us = rounds * (a + c * [ksplit > 1] + d * max(0, ksplit - 2) + b * stages_per_workgroup),    rounds = ceil(workgroups / 256)

by least squares on the relative error (a: launch, fill, epilogue; c, d: the in-launch split-K hand-off; b: microseconds per 128-k stage).  The K split of
each measured point is the one the library plans for that forced variant (`qqq_w4a8_plan`; it does not depend on this table).  A kernel change that moves
these rates is followed by: re-measure (tools/visits/r5_v07.sh), re-run this tool, rebuild -- no constant is edited by hand.

The wide kernel's three tile shapes (256 x 256, 256 x 128, 128 x 256; columns wide / w16x2 / w128 / w128x2 / w8 of the checks run with WIDE_SHAPES=1) get the
three rates of wide_estimate's form -- us = 3.7 + rounds * (fixed + handoff * [ksplit > 1] * tile KiB / 256 + stages per workgroup * t_stage * load) with the
rounds / load rules of qqq_w4a8.hip -- fitted the same way per (shape, mode).

Up to 64 tokens the column kernel (1 ... 32 tokens) and the stream kernel get one linear form per token range and mode (small_features below; columns column /
stream of the checks), fitted the same way.

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


W8_FILE = os.path.join(ROOT, "profiles", "r06_w8_dispatch_check_main.txt")  # the wide shapes on EXPANDED int8 weights (mode g128x): the third column of kQqqWideRates


def collect_wide(files):
    """{(shape, False | True | "w8"): [(features, measured us)]} -- per-channel / per-group from the grids, "w8" (calls that have the layer's expanded weights) from W8_FILE"""
    import dispatch_regret as R
    from qqq_amd import _lib

    data = {}
    for f in list(files) + [W8_FILE]:
        for line in open(f):
            m = R.LINE.match(line)
            if not m or "w16x2" not in line:
                continue
            N, K, mode, M = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
            if K % 128 or (mode == "g128x") != (f == W8_FILE):
                continue
            cells = m.group(8).split("<--")[0].split()
            meas = {cells[i]: float(cells[i + 1]) for i in range(0, len(cells) - 1, 2)}
            for col, tune in WIDE_COLS.items():
                v = meas.get(col)
                if v is None or v != v:
                    continue
                w8 = mode == "g128x"
                p = _lib.plan(M, N, K, -1 if mode == "pc" else 128, 16, tune=dict(tune, w8=1) if w8 else tune)
                if p["kernel"] != 5 or p["glds"] == 2 or (col.endswith("x2") and p["ksplit"] != 2) or (w8 and not p["w8"]):
                    continue
                rows, bn, ks = 16 * p["mt"], p["bm"], p["ksplit"]
                tl = ((M + rows - 1) // rows) * ((N + bn - 1) // bn)
                shape = 0 if (p["mt"] == 16 and bn == 256) else 1 if p["mt"] == 16 else 2
                # (expanded weights: wide_estimate prices the call like a per-channel one -- the loop has no re-quantiser -- with this column's rates)
                data.setdefault((shape, "w8" if w8 else mode == "g128"), []).append((wide_features(tl, ks, rows, bn, K // 128, mode == "g128"), v))
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
    from scipy.optimize import nnls

    coef = nnls(A * w[:, None], y * w)[0]  # no negative rates: with two or three distinct K splits in a group a free fit buys its last per cent with a hand-off that gets cheaper per slice
    if not A[:, 1].any() or (A[:, 1] == A[:, 0]).all():  # the group was only ever measured split (or only unsplit): a and c are not separable -- share the sum
        s = coef[0] + coef[1]
        coef[0], coef[1] = 0.75 * s, 0.25 * s
    e = (A @ coef) / y - 1.0
    return coef, len(y), float(np.abs(e).mean()), float(e.mean()), float(np.abs(e).max())


def collect_small(files):
    """forced column (up to 32 tokens) / stream (up to 256) measurements: [(family, per-group, M, N, K, us)]"""
    import dispatch_regret as R

    out = []
    for f in files:
        for line in open(f):
            m = R.LINE.match(line)
            if not m:
                continue
            N, K, mode, M = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
            if M > 256:
                continue
            cells = m.group(8).split("<--")[0].split()
            meas = {cells[i]: float(cells[i + 1]) for i in range(0, len(cells) - 1, 2)}
            for fam in ("column", "stream"):
                v = meas.get(fam)
                if v is not None and v == v and (M <= 64 or fam == "stream"):
                    out.append((fam, mode == "g128", M, N, K, v))
    return out


def small_features(fam, grouped, M, N, K):
    """the linear forms of column_small_estimate / stream_small_estimate / the 33 ... 64-token branch of stream_estimate (qqq_w4a8.hip); None outside a form's range"""
    if fam == "column":
        wgs = N // 32
        rounds = (wgs + 255) // 256
        if M > 32 or wgs < 16 or wgs > 768:
            return None
        if not grouped:  # launch + weights + the activations every 32-column workgroup reads again + the serial depth of a chip that is not full
            return "col_pc", [1.0, N * K / 2e6, K * M * rounds / 1e6, K / 1000.0 * max(0.0, 1.0 - wgs / 256.0)]
        r = 1.0 if rounds == 1 else 0.92 * rounds  # per-group: the re-quantiser (time ~ K per round), a second 16-token tile, the weights
        return "col_g", [1.0, K * r / 1000.0, (K * r / 1000.0) * (((M - 16) / 16.0) ** 0.75 if M > 16 else 0.0), N * K / 2e6]
    pb = N * K / 2.0 / 5.0e6  # one pass over the weights at 5 TB/s, us
    if M <= 16:
        return "st16", [1.0, pb]
    if M <= 32:
        return "st32", [1.0, (M - 24) / 16.0, pb]
    if M <= 64:
        four = 1.0 if M > 48 else 0.0
        return "st64", [1.0, float(M), four, pb, pb * M, pb * four]
    # 65 ... 256 tokens (two to four 64-token m-blocks) on layers of up to ~40 MB: stream_mid_estimate -- launch + the loop (per 64-k step of a slice, per round of 256
    # workgroups, x 1.2 from the second round on) + for a K split the slabs and the reduce launch; the split is the one the library plans for the forced stream kernel
    mb = (M + 63) // 64
    if mb > 4 or pb >= 8.0:
        return None
    from qqq_amd import _lib

    ks = _lib.plan(M, N, K, 128 if grouped else -1, 16, tune=dict(kernel=1))["ksplit"]
    rounds = (((N + 127) // 128) * mb * ks + 255) // 256
    return "stmid", [1.0, rounds * (1.2 if rounds > 1 else 1.0) * (K // 64) / ks, 1.0 if ks > 1 else 0.0, M * N * 4.0 * ks / 1e6 if ks > 1 else 0.0]


def fit_small(files):
    from scipy.optimize import nnls

    groups = {}
    for fam, g, M, N, K, v in collect_small(files):
        ft = small_features(fam, g, M, N, K)
        if ft is None:
            continue
        key = (ft[0], g) if fam == "stream" else (ft[0], ft[0] == "col_g")
        groups.setdefault(key, []).append((ft[1], v))
    res = {}
    for key, rows in sorted(groups.items()):
        A = np.array([r[0] for r in rows], float)
        y = np.array([r[1] for r in rows], float)
        w = 1.0 / y
        coef = nnls(A * w[:, None], y * w)[0] if key[0] == "stmid" else np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
        e = (A @ coef) / y - 1.0
        res[key] = (coef, len(y), float(np.abs(e).mean()), float(e.mean()), float(np.abs(e).max()))
        print(f"{key[0]:7s} {'g128' if key[1] else 'pc  '} n={len(y):4d}  " + " ".join(f"{c:9.4f}" for c in coef) + f"   mean |err| {100 * res[key][2]:4.1f}%  bias {100 * res[key][3]:+4.1f}%  worst {100 * res[key][4]:4.1f}%")
    return res


def fit_panel64(files):
    """several 64-token m-blocks above 64 tokens in ONE round of workgroups (column panel64 of the checks): us = (f1 unsplit | f2 split) + max(stages per slice x st,
    (1 + extra x (m-blocks - 1)) weight passes at bw TB/s) -- the max makes it a non-linear fit (scipy least_squares on the relative error); per mode"""
    import dispatch_regret as R
    from qqq_amd import _lib
    from scipy.optimize import least_squares

    pts = {False: [], True: []}
    for f in files:
        for line in open(f):
            m = R.LINE.match(line)
            if not m:
                continue
            N, K, mode, M = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
            cells = m.group(8).split("<--")[0].split()
            meas = {cells[i]: float(cells[i + 1]) for i in range(0, len(cells) - 1, 2)}
            v = meas.get("panel64")
            if v is None or v != v:
                continue
            g = mode == "g128"
            ks = _lib.plan(M, N, K, 128 if g else -1, 16, tune=dict(kernel=4, mt=4))["ksplit"]
            mb4 = (M + 63) // 64
            if mb4 * ((N + 127) // 128) * ks > 256:
                continue
            pts[g].append((((K // 64 + 1) // 2) / ks, ks, mb4, N * K / 2.0e6, v))
    out = {}
    for g in (False, True):
        P = pts[g]

        def model(par, q):
            return (par[0] if q[1] == 1 else par[1]) + max(q[0] * par[2], (1.0 + par[3] * (q[2] - 1)) * q[3] / par[4])

        r = least_squares(lambda par: [model(par, q) / q[4] - 1.0 for q in P], [8.4, 10.4, 0.52, 0.8, 3.66] if g else [8.0, 10.0, 0.316, 0.8, 5.3],
                          bounds=([0, 0, 0, 0, 1.0], [30, 30, 5, 2, 8.0]))
        e = np.array([model(r.x, q) / q[4] - 1.0 for q in P])
        out[g] = (r.x, len(P), float(np.abs(e).mean()), float(e.mean()), float(np.abs(e).max()))
        print(f"panel64 {'g128' if g else 'pc  '} n={len(P):4d}  " + " ".join(f"{c:8.4f}" for c in r.x) + f"   mean |err| {100 * out[g][2]:4.1f}%  bias {100 * out[g][3]:+4.1f}%  worst {100 * out[g][4]:4.1f}%")
    return out


def main():
    import dispatch_regret as R

    files = R.grid_files()  # one file per grid: round 6's measurement where there is one (the wide kernel's grids), else round 5's
    data = collect(files)
    names = ("128-column strips, 32 columns per wave", "256-column strips, 32 columns per wave", "256-column strips, 64 columns per wave (128-token m-blocks only)")
    lines = ["// qqq_rates.h -- GENERATED by tools/fit_rates.py from the newest profiles/r0N_dispatch_check_<grid>.txt of every grid (%d files); do not edit by hand." % len(files),
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
              "struct QqqWideRate { double fixed, handoff, t_stage; };", "// [shape][per-channel, per-group, expanded int8 weights (profiles/r06_w8_dispatch_check_main.txt)]",
              "static const QqqWideRate kQqqWideRates[3][3] = {"]
    fitted = {}
    for shape in range(3):
        for g in (False, True, "w8"):
            fitted[(shape, g)] = fit_wide(wide[(shape, g)])
    label = {False: "per-channel", True: "per-group", "w8": "expanded"}
    for shape in range(3):
        cells = []
        for g in (False, True, "w8"):
            coef, n, mae, bias, worst = fitted[(shape, g)]
            if coef[1] == 0.0:
                coef = np.array([coef[0], fitted[(1, g)][0][1], coef[2]])  # (never measured split)
            cells.append("{%.3f, %.3f, %.4f}  /* %s: %d points, %.1f %%, %.0f %% */" % (coef[0], coef[1], coef[2], label[g], n, 100 * mae, 100 * worst))
            print(f"wide shape {shape} {label[g]:11s} n={n:4d}  fixed={coef[0]:6.2f} handoff={coef[1]:5.2f} t_stage={coef[2]:.4f}   mean |err| {100 * mae:4.1f}%  bias {100 * bias:+4.1f}%  worst {100 * worst:4.1f}%")
        lines.append("    {%s,\n     %s,\n     %s},  // %s" % (cells[0], cells[1], cells[2], wnames[shape]))
    lines += ["};", ""]
    small = fit_small(files)

    p64 = fit_panel64(files)

    def arr64(g):
        coef, n, mae, bias, worst = p64[g]
        return "{" + ", ".join("%.5g" % c for c in coef) + "}  /* %d points, %.1f %%, %.0f %% */" % (n, 100 * mae, 100 * worst)

    def arr(key):
        coef, n, mae, bias, worst = small[key]
        return "{" + ", ".join("%.5g" % c for c in coef) + "}  /* %d points, %.1f %%, %.0f %% */" % (n, 100 * mae, 100 * worst)

    lines += ["// up to 64 tokens: the column kernel (1 ... 32 tokens) and the stream kernel, one linear form per kernel, token range and mode (column_small_estimate, stream_small_estimate,",
              "// stream_estimate in qqq_w4a8.hip say what the terms are); MB = N K / 2e6, pb = a pass over the weights at 5 TB/s in us, r = rounds of 256 workgroups (x 0.92 from the second)",
              "struct QqqSmallRates {",
              "  double col_pc[4];   // c0 + c1 MB + c2 K m rounds / 1e6 + c3 K / 1000 max(0, 1 - workgroups / 256)",
              "  double col_g[4];    // g0 + g1 K r / 1000 + g2 (K r / 1000) ((m - 16) / 16)^0.75 [m > 16] + g3 MB",
              "  double st16[2][2];  // [per-group]: a + b pb                                                    (up to 16 tokens)",
              "  double st32[2][3];  // a + c (m - 24) / 16 + b pb                                               (17 ... 32)",
              "  double st64[2][6];  // f0 + f1 m + f2 [m > 48] + pb (p0 + p1 m + p2 [m > 48])                   (33 ... 64)",
              "  double stmid[2][4]; // a + b steps per slice x rounds (x 1.2 from the second) + [split] (s0 + s1 MB of int32 slabs) (65 ... 256, layers up to ~40 MB)",
              "  double panel64[2][5]; // panel kernel, several 64-token m-blocks above 64 tokens, one round: (f1 unsplit | f2 split) + max(stages per slice x st, (1 + extra (m-blocks - 1)) MB / bw), bw in MB/us",
              "};",
              "static const QqqSmallRates kQqqSmall = {",
              "    " + arr(("col_pc", False)) + ",", "    " + arr(("col_g", True)) + ",",
              "    {" + arr(("st16", False)) + ",\n     " + arr(("st16", True)) + "},",
              "    {" + arr(("st32", False)) + ",\n     " + arr(("st32", True)) + "},",
              "    {" + arr(("st64", False)) + ",\n     " + arr(("st64", True)) + "},",
              "    {" + arr(("stmid", False)) + ",\n     " + arr(("stmid", True)) + "},",
              "    {" + arr64(False) + ",\n     " + arr64(True) + "},",
              "};", "", "#endif  // QQQ_AMD_QQQ_RATES_H_", ""]
    open(OUT, "w").write("\n".join(lines))
    print("wrote", os.path.relpath(OUT, ROOT))


if __name__ == "__main__":
    main()
