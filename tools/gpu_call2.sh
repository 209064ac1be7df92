#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c2; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden_fixtures or random_shapes or fused_bias" > $O/pytest_panel_small.log 2>&1; echo "rc=$?" >> $O/pytest_panel_small.log
tail -n 25 $O/pytest_panel_small.log
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 12 $O/pytest_all.log
timeout 900 python tools/panel_sweep.py > $O/panel_sweep.txt 2>&1
tail -n 70 $O/panel_sweep.txt
