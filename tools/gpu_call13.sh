#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c13; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 5 $O/pytest_all.log
MODES=pc MS=128,256,1024,4096 TUNES="[dict(kernel=2), dict(kernel=4,mt=8,bm=128), dict(kernel=4,mt=8,bm=128,glds=2), dict(kernel=4,mt=8,bm=256,pf=3), dict(kernel=4,mt=8,bm=256,glds=2)]" timeout 900 python tools/big_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sk_sweep.txt; cat $O/sk_sweep.txt
MODES=g128 MS=256,1024 TUNES="[dict(kernel=2), dict(kernel=4,mt=8,bm=128), dict(kernel=4,mt=8,bm=128,glds=2), dict(kernel=4,mt=8,bm=256,pf=3), dict(kernel=4,mt=8,bm=256,glds=2)]" timeout 900 python tools/big_sweep.py 2>&1 | grep -v amdgpu.ids >> $O/sk_sweep.txt; tail -10 $O/sk_sweep.txt
