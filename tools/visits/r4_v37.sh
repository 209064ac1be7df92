#!/bin/bash
# round 4, visit 37: larger remainders for the M split (QQQ_AMD_SPLIT_CAP=4096 against the shipped 512): does the model's choice hold when the remainder is itself a big call?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v37; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
for cap in 512 4096; do
  echo "## QQQ_AMD_SPLIT_CAP=$cap" | tee -a $O/cap.txt
  QQQ_AMD_SPLIT_CAP=$cap STEP=100000 MS=4700,5000,5400,5900,6500,7000,9000,10000,3000,2600,1300,1500 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/cap.txt
  QQQ_AMD_SPLIT_CAP=$cap STEP=100000 NK=4096,4096 NBUF=12 MS=9000,10000,11000,13000,5000,6000,3000 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/cap.txt
  QQQ_AMD_SPLIT_CAP=$cap STEP=100000 NK=11008,4096 NBUF=12 MS=9000,10000,5000,3000 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/cap.txt
  QQQ_AMD_SPLIT_CAP=$cap STEP=100000 NK=4096,11008 NBUF=12 MS=9000,10000,5000,6000,3000 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/cap.txt
  QQQ_AMD_SPLIT_CAP=$cap STEP=100000 MODE=g128 MS=5000,6000,3000 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/cap.txt
done
