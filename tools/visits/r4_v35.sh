#!/bin/bash
# round 4, visit 35: what a ragged token count costs (one more m-block of 256 x 256 tiles = a partial extra round) and what an M split would save
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v35; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
MS=4096,4097,4100,4160,4224,4352,2049,2100,8200,1025,1100 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged.txt
MODE=g128 MS=4097,4224,1030 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged.txt
NK=4096,4096 NBUF=12 MS=8193,8200,8320,4100,2050,1030 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged.txt
NK=11008,4096 NBUF=12 MS=8200,4100,1030 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged.txt
NK=4096,11008 NBUF=12 MS=8200,4100 timeout 600 python tools/ragged_m.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ragged.txt
