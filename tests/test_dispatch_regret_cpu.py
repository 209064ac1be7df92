"""The dispatcher against this repository's own hardware measurements, without a GPU: every committed `tools/dispatch_check.py`
output of round 5 (automatic choice and every forced family -- column, stream, the panel kernel's strip shapes and m-blocks, every wide shape -- timed
interleaved on an MI355X running the round's FINAL library, cold -- 1.1 GB of rotating weight copies per layer -- and with the variants of a point in a different order every round; the grids are round 4's)
is replayed against the plans of the CURRENT library (`qqq_w4a8_plan` is pure host logic).  A change to a cost model that sends some measured point to a
clearly slower family fails here; the bounds are what the final library of round 5 reaches on these files plus a little room (the measurements carry their
box's noise: 3-5 % at the 10-20 us points).  How far each model is from the clock in absolute terms: tests/test_cost_models_cpu.py."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# grid -> (least number of points, most points above 3 %, worst regret allowed).  Round 6: the grids that price the wide kernel (and panel64, which times it above 256 tokens) were
# measured again on the round's library -- its loop is 4 - 7 % faster -- and tools/dispatch_regret.grid_files() hands out the newest measurement of every grid.
FILES = {
    "m16": (95, 10, 0.15),  # ten shapes at 9 ... 32 tokens (round 5)
    "m64": (76, 4, 0.08),  # ten shapes at 40 ... 64 tokens (round 5)
    "main": (76, 8, 0.10),  # BASELINE + Llama-2-7B layers, 1 ... 8192 tokens, every wide shape
    "merged": (90, 5, 0.10),  # merged projections (N = 12288 / 22016) and narrow layers (N = 2048 / 1024)
    "mid": (16, 2, 0.04),  # BASELINE layer at 320 ... 3072 tokens
    "mid_shapes": (135, 8, 0.10),  # nine shapes at 96 ... 4096 tokens, every wide shape
    "more_models": (150, 12, 0.13),  # Yi-34B / Phi-3 / Llama-70B k,v / Qwen2-72B layers
    "panel64": (205, 14, 0.10),  # fourteen shapes at 80 ... 512 tokens with the 64-token m-block column
    "qwen_mistral": (132, 11, 0.18),  # Qwen2-7B / Mistral-7B layers
    "shapes": (80, 5, 0.08),  # six other layer shapes
}


def _grid_file(name):
    import re

    import dispatch_regret as R

    for f in R.grid_files():
        if re.fullmatch(rf"r\d+_dispatch_check_{name}\.txt", os.path.basename(f)):
            return f
    raise AssertionError(name)


@pytest.mark.parametrize("name", sorted(FILES))
def test_planned_family_is_close_to_the_best_measured_one(name):
    import dispatch_regret as R

    rows = R.regrets(_grid_file(name))
    n, above, worst = R.summary(rows)
    need, most, bound = FILES[name]
    bad = sorted(rows, key=lambda r: r[6] / r[5])[:5]
    assert n >= need, (name, n)
    assert above <= most and worst <= bound, (name, n, above, worst, bad)
