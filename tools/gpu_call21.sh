#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c21; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 6 $O/pytest_all.log
timeout 900 python tools/decode_sweep.py 2>&1 | grep -v amdgpu.ids > $O/decode_sweep.txt; cat $O/decode_sweep.txt
