#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c9; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 5 $O/pytest_all.log
MS=48,64,96,128,160,192,256,320,384,512,768,1024 timeout 1200 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_b.txt; grep -c "<--" $O/dispatch_check_b.txt; grep "<--" $O/dispatch_check_b.txt
