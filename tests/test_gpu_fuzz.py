"""A short differential fuzz inside the GPU suite (tools/fuzz_families.py runs the same loop for minutes): random valid problems
x random tuning variants and the automatic dispatch, every result bit-identical to the unsplit stream kernel (which the parity
tests pin against the CPU oracle), half of the launches on a loaded chip."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_differential_fuzz(seed):
    env = dict(os.environ, SEED=str(seed), SECONDS="6")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_families.py")], env=env, capture_output=True, text=True, timeout=300)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-2000:]
    assert "MISMATCH" not in out and "ERROR" not in out, out[-2000:]
    assert f"seed {seed}: done" in out, out[-2000:]


def test_tile_walk_differential_fuzz():
    """tools/fuzz_walk.py: random large problems (1 ... 4 tiles per workgroup of the walk's grid, K = 8 ... 48 stages, ragged edges,
    bias on / off, idle and loaded chip) through the persistent tile walk in its three tile shapes and through the automatic
    dispatch, bit-identical to the tiled kernel."""
    env = dict(os.environ, SEED="21", SECONDS="12")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_walk.py")], env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-2000:]
    assert "MISMATCH" not in out and "ERROR" not in out, out[-2000:]
    assert "seed 21: done" in out, out[-2000:]
    walked = int(out.split(" of them the tile walk")[0].rsplit(" ", 1)[1])
    assert walked >= 3, out[-500:]
