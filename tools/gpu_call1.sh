#!/bin/bash
# round-2 GPU visit 1: full -m gpu suite (new coverage tests included) + smoke + bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2c1
export TMPDIR=/tmp
O=gpurun_out/r2c1
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > $O/box.txt 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 45 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err; head -c 1500 $O/bench.json
