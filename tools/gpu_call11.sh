#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c11; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 5 $O/pytest_all.log
MODES=pc,g128 timeout 900 python tools/big_sweep.py 2>&1 | grep -v amdgpu.ids > $O/big_sweep.txt; cat $O/big_sweep.txt
