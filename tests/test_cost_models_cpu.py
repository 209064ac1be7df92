"""The dispatcher's cost models against the clock, without a GPU: tools/cost_model_report.py asks the CURRENT library for its price of every kernel family
(`qqq_w4a8_model_us`, pure host logic) at every point of the committed dispatch checks (round 5: 1114 points on ten grids, every family forced and timed
on an MI355X running the round's final library, with 1.1 GB of rotating weight copies per layer so that no cache serves the small token counts) and compares it with the fastest measured variant of that family.  The bounds are what the library of round 5 reaches (profiles/r05_cost_model_error.txt)
plus room for the measurements' own box-to-box spread; a kernel or model change that moves a family's price away from the clock fails here and says where.
(That the models ORDER the families correctly is tests/test_dispatch_regret_cpu.py's business.)"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# (family, token regime) -> (least points, largest mean |error|, largest |bias|)
BOUNDS = {
    # every family is priced from GENERATED rates (tools/fit_rates.py -> qqq_amd/csrc/qqq_rates.h) at the end of round 5: panel + wide tables, and up to 256 tokens the
    # linear forms of the column / stream kernels (the stream kernel's large-layer branch above 64 tokens was the one hand-fitted piece left; round 6 took that regime out of the
    # automatic path -- never chosen, never more than 3 % ahead in its 48 measured points -- and it is priced for the report only)
    ("column", "1-8"): (85, 0.05, 0.02), ("column", "9-32"): (180, 0.05, 0.02),
    ("stream", "1-8"): (85, 0.055, 0.03), ("stream", "9-32"): (180, 0.05, 0.02), ("stream", "33-64"): (125, 0.045, 0.02), ("stream", "65-256"): (300, 0.06, 0.02),
    ("panel", "9-32"): (180, 0.045, 0.025), ("panel", "33-64"): (125, 0.04, 0.02), ("panel", "65-256"): (300, 0.045, 0.035), ("panel", "257-1024"): (230, 0.05, 0.03),
    ("panel", ">1024"): (155, 0.055, 0.03), ("wide", "257-1024"): (150, 0.04, 0.03), ("wide", ">1024"): (155, 0.035, 0.02),
}


def test_every_model_is_within_its_bound_of_the_measurements():
    import cost_model_report as C

    import dispatch_regret

    files = dispatch_regret.grid_files()  # (round 6: the grids that price the wide kernel were measured again on its new loop)
    assert len(files) >= 10
    t = C.table([r for f in files for r in C.points(f)])
    assert set(BOUNDS) <= set(t), sorted(set(BOUNDS) - set(t))
    for key, (need, mae, bias) in BOUNDS.items():
        n, got_mae, got_bias, worst, point = t[key]
        assert n >= need and got_mae <= mae and abs(got_bias) <= bias, (key, n, got_mae, got_bias, point)


def test_model_prices_follow_the_plan():
    """the family the plan picks is the cheapest of the prices the report sees (same functions, same arguments)"""
    from qqq_amd import _lib

    fam = {1: "stream", 3: "column", 4: "panel", 5: "wide"}
    for gs in (-1, 128):
        for (n, k) in ((8192, 21760), (4096, 4096), (11008, 4096), (4096, 11008)):
            for m in (40, 64, 100, 128, 200, 256, 300, 512, 1024, 4096):
                prices = _lib.model_us(m, n, k, gs, 16)
                plan = _lib.plan(m, n, k, gs, 16, tune=dict(split_m=-1))
                best = min(prices, key=prices.get)
                assert fam[plan["kernel"]] == best or abs(prices[fam[plan["kernel"]]] - prices[best]) < 1e-9, (m, n, k, gs, prices, plan["kernel"])


def test_rate_tables_are_what_the_tool_generates():
    """qqq_amd/csrc/qqq_rates.h is GENERATED (tools/fit_rates.py, least squares over every forced panel / wide / column / stream variant of the committed dispatch checks):
    re-running the fit on the committed measurements reproduces the committed coefficients -- nobody edited a table by hand, and nobody changed the
    measurements without regenerating them."""
    import re

    import fit_rates as F

    import dispatch_regret

    files = dispatch_regret.grid_files()  # (round 6: the grids that price the wide kernel were measured again on its new loop)
    text = open(F.OUT).read()
    panel_text, wide_text = text.split("struct QqqWideRate")
    wide_text, small_text = wide_text.split("struct QqqSmallRates")
    rows = [tuple(float(v) for v in m) for m in re.findall(r"\{(-?[\d.]+), (-?[\d.]+), (-?[\d.]+), (-?[\d.]+)\}", panel_text)]
    assert len(rows) == 24
    data = F.collect(files)
    i = 0
    fitted = 0
    for ci in range(3):
        for mt in F.MTS:
            for g in (False, True):
                pts = data.get((ci, mt, g))
                if pts and len(pts) >= 8:
                    coef, n, mae, bias, worst = F.fit(pts)
                    assert all(abs(a - b) < 2e-3 for a, b in zip(coef, rows[i])), (ci, mt, g, coef, rows[i])
                    assert mae < 0.06, (ci, mt, g, mae)
                    fitted += 1
                else:
                    assert rows[i] == (0.0, 0.0, 0.0, 0.0)
                i += 1
    assert fitted == 18
    wrows = [tuple(float(v) for v in m) for m in re.findall(r"\{(-?[\d.]+), (-?[\d.]+), (-?[\d.]+)\}", wide_text)]
    assert len(wrows) == 9  # (round 6: a third column, calls that have the layer's expanded int8 weights -- one grid of four layers)
    wide = F.collect_wide(files)
    for shape in range(3):
        for gi, g in enumerate((False, True, "w8")):
            coef, n, mae, bias, worst = F.fit_wide(wide[(shape, g)])
            got = wrows[3 * shape + gi]
            assert n >= (25 if g == "w8" else 150) and mae < (0.10 if g == "w8" else 0.05) and abs(coef[0] - got[0]) < 2e-3 and abs(coef[2] - got[2]) < 2e-4, (shape, g, coef, got, mae)
            if coef[1] != 0.0:
                assert abs(coef[1] - got[1]) < 2e-3
    # the small-m forms of the column / stream kernels and the 64-token m-block form of the panel kernel (kQqqSmall), in the order of the initialiser
    init = small_text.split("kQqqSmall = {")[1]
    got = [[float(v) for v in re.findall(r"-?[\d.]+(?:e-?\d+)?", grp)] for grp in re.findall(r"\{([^{}]*)\}\s*/\*", init)]
    small = F.fit_small(files)
    p64 = F.fit_panel64(files)
    want = [small[("col_pc", False)], small[("col_g", True)]] + [small[(k, g)] for k in ("st16", "st32", "st64", "stmid") for g in (False, True)] + [p64[False], p64[True]]
    assert len(got) == len(want) == 12
    for (coef, n, mae, bias, worst), g in zip(want, got):
        assert len(coef) == len(g) and all(abs(a - b) <= 2e-4 * max(1.0, abs(a)) for a, b in zip(coef, g)), (coef, g)
        assert n >= 38 and mae < 0.05, (n, mae)
