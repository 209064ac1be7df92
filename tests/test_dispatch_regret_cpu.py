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

# file -> (least number of points, most points above 3 %, worst regret allowed)
FILES = {
    "r05_dispatch_check_m16.txt": (95, 10, 0.15),  # ten shapes at 9 ... 32 tokens
    "r05_dispatch_check_m64.txt": (76, 4, 0.08),  # ten shapes at 40 ... 64 tokens
    "r05_dispatch_check_main.txt": (76, 8, 0.10),  # BASELINE + Llama-2-7B layers, 1 ... 8192 tokens, every wide shape
    "r05_dispatch_check_merged.txt": (90, 4, 0.10),  # merged projections (N = 12288 / 22016) and narrow layers (N = 2048 / 1024)
    "r05_dispatch_check_mid.txt": (16, 2, 0.04),  # BASELINE layer at 320 ... 3072 tokens
    "r05_dispatch_check_mid_shapes.txt": (135, 8, 0.10),  # nine shapes at 96 ... 4096 tokens, every wide shape
    "r05_dispatch_check_more_models.txt": (150, 9, 0.13),  # Yi-34B / Phi-3 / Llama-70B k,v / Qwen2-72B layers
    "r05_dispatch_check_panel64.txt": (205, 14, 0.09),  # fourteen shapes at 80 ... 512 tokens with the 64-token m-block column
    "r05_dispatch_check_qwen_mistral.txt": (132, 11, 0.18),  # Qwen2-7B / Mistral-7B layers
    "r05_dispatch_check_shapes.txt": (80, 5, 0.07),  # six other layer shapes
}


@pytest.mark.parametrize("name", sorted(FILES))
def test_planned_family_is_close_to_the_best_measured_one(name):
    import dispatch_regret as R

    rows = R.regrets(os.path.join(ROOT, "profiles", name))
    n, above, worst = R.summary(rows)
    need, most, bound = FILES[name]
    bad = sorted(rows, key=lambda r: r[6] / r[5])[:5]
    assert n >= need, (name, n)
    assert above <= most and worst <= bound, (name, n, above, worst, bad)
