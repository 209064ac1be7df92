#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 12 $O/pytest_all.log
MODES=g128 timeout 900 python tools/panel_sweep.py > $O/panel_sweep_g128.txt 2>&1
grep -A4 "^==" $O/panel_sweep_g128.txt | head -60; grep "auto:" $O/panel_sweep_g128.txt
