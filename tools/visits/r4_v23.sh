#!/bin/bash
# round 4, visit 23: 128 x 256 tiles (w8) against 256 x 128 tiles (w128) of the wide kernel at 640 ... 1536 tokens, cold weights (5 / 12 rotating buffers)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v23; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=5,mt=8), dict(kernel=5,bm=128), dict(kernel=5,bm=128,ksplit=2)]"
run() { NK=$1 MS=$2 MODE=$3 NBUF=$4 ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$1 $3 /" | tee -a $O/ab.txt; }
run 8192,21760 640,768,1024,1280 pc 5
run 8192,8192 768,1024,1536 pc 12
run 4096,11008 1024,1536,2048 pc 12
run 5120,13824 1024,1536 pc 12
run 4096,4096 1536,2048 pc 12
run 11008,4096 768,1024,1536 pc 12
