#!/bin/bash
# round 4, visit 21: walk vs one tile per workgroup at K = 8192 / 11008 / 21760 on (yet) another box: where should the automatic rule stop?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v21; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[dict(kernel=5,glds=1), dict(kernel=5,glds=2)]"
run() { NK=$1 MS=$2 MODE=$3 NBUF=${4:-8} ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$1 $3 /" | tee -a $O/ab.txt; }
run 8192,8192 2048,4096,8192,16384 pc
run 8192,8192 4096,8192 g128
run 28672,8192 1024,2048,4096 pc 4
run 6144,6144 4096,8192 pc
run 7168,7168 4096,8192 pc
run 4096,11008 8192,16384 pc
run 8192,21760 4096,8192 pc 4
run 8192,21760 4096 g128 4
