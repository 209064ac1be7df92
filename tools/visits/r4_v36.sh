#!/bin/bash
# round 4, visit 36: the automatic M split -- parity test, the ragged-token-count table again (one call = automatic, now split where the models say so), whole suite, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v36; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "m_split" 2>&1 | tail -5 | tee $O/pytest_split.log
MS=4096,4097,4100,4160,4224,4352,2049,2100,8200,1025,1100,5000 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged_after.txt
MODE=g128 MS=4097,4224,1030 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged_after.txt
NK=4096,4096 NBUF=12 MS=8193,8200,8320,4100,2050,1030 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged_after.txt
NK=11008,4096 NBUF=12 MS=8200,4100,1030 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged_after.txt
NK=4096,11008 NBUF=12 MS=8200,4100 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged_after.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print('value', round(d['value'],1), [round(v['us_median'],1) for v in d['per_m'].values()])"
