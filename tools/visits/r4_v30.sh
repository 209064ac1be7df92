#!/bin/bash
# round 4, visit 30: stream K split one short of a 257th workgroup (4-wave bodies too?) on 7168 x 20480; per-group decode on 20480 x 7168
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v30; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=3), dict(kernel=1,ksplit=3), dict(kernel=1,ksplit=4), dict(kernel=1,ksplit=5)]"
run() { NK=$1 MS=$2 MODE=$3 NBUF=${4:-6} ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$1 $3 /" | tee -a $O/ab.txt; }
run 7168,20480 1,16 pc
run 7168,20480 1,16 g128
run 11008,4096 1,16 pc 12
run 5120,13824 16 pc 12
T="[None, dict(kernel=3), dict(kernel=1,ksplit=1), dict(kernel=1,ksplit=2), dict(kernel=3,mt=1,pf=8)]"
run 20480,7168 1,16 g128
run 20480,7168 1,16 pc
