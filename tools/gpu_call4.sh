#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 12 $O/pytest_all.log
MODES=pc timeout 900 python tools/panel_sweep.py > $O/panel_sweep_pc.txt 2>&1
grep -A6 "^==" $O/panel_sweep_pc.txt | head -60; grep "auto:" $O/panel_sweep_pc.txt
