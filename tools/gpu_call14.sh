#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c14; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 5 $O/pytest_all.log
ROUNDS=4 ITERS=4 MS=4096,2048 TUNES="[dict(kernel=2,bm=256,glds=1,stages=6), dict(kernel=2,bm=256,glds=1,stages=7)]" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_ns7.txt
