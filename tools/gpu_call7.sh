#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c7; mkdir -p $O
export TMPDIR=/tmp
MODES=pc MS=48,64,96,128,160,192,256,320 timeout 900 python tools/panel_sweep.py > $O/panel_sweep_pc.txt 2>&1
grep -A4 "^==" $O/panel_sweep_pc.txt | head -80; grep "auto:" $O/panel_sweep_pc.txt
MODES=g128 MS=16,32,64,128,192,256 timeout 900 python tools/panel_sweep.py > $O/panel_sweep_g128.txt 2>&1
grep -A4 "^==" $O/panel_sweep_g128.txt | head -80; grep "auto:" $O/panel_sweep_g128.txt
