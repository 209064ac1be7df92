#!/bin/bash
# round 5, visit 37: 1024 tokens per-channel as 256 x 256 tiles in two K slices (the plan since the last refit): the uneven slices' skew around the automatic 8
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v37; mkdir -p gpurun_out/r5v37; export TMPDIR=/tmp
T="[None, dict(kernel=5,ksplit=2,skew=4), dict(kernel=5,ksplit=2,skew=6), dict(kernel=5,ksplit=2,skew=10), dict(kernel=5,ksplit=2,skew=12), dict(kernel=5,ksplit=2,skew=16), dict(kernel=5,ksplit=2,pw=4), dict(kernel=5,ksplit=2,pw=16), None, dict(kernel=5,ksplit=2,skew=12)]"
NBUF=5 ROUNDS=8 ITERS=4 MS=1024,896 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
