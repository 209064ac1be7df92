#!/bin/bash
# round 5, visit 36: 128 tokens on the BASELINE layer in 5 / 6 / 8 K slices of 128-column strips (two workgroups per CU from 5 slices on): does a second resident workgroup
# close the gap between the loop (0.557 us per stage) and its two floors (memory path 0.46-0.48, instructions 0.475)?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v36; mkdir -p gpurun_out/r5v36; export TMPDIR=/tmp
timeout 200 python tools/check_variant.py --ms 128 --tunes "[dict(kernel=4,ksplit=8), dict(kernel=4,ksplit=6), dict(kernel=4,ksplit=8,skew=2)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/check.txt
T="[None, dict(kernel=4,ksplit=5), dict(kernel=4,ksplit=6), dict(kernel=4,ksplit=8), dict(kernel=4,ksplit=8,skew=-1), dict(kernel=4,ksplit=8,skew=2), dict(kernel=4,ksplit=8,skew=4), None, dict(kernel=4,ksplit=8)]"
NBUF=0 ROUNDS=8 ITERS=4 MS=128,96 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 NBUF=0 ROUNDS=8 ITERS=4 MS=128 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
