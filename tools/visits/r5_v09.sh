#!/bin/bash
# round 5, visit 9: differential fuzz soak of the final library (random problems x random variants incl. random skews, idle and loaded chip; the tile walk)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v09; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
for s in 501 502 503 504 505 506; do SEED=$s SECONDS=50 timeout 200 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/fuzz.txt; done
for s in 521 522; do SEED=$s SECONDS=50 timeout 300 python tools/fuzz_walk.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/fuzz.txt; done
