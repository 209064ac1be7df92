#!/bin/bash
# round 4, visit 28: the stream kernel's K split at 9 ... 16 tokens when its strips alone are more than half a round (128 < N / 128 < 256)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v28; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=3), dict(kernel=1,ksplit=1), dict(kernel=1,ksplit=2), dict(kernel=1,ksplit=3)]"
run() { NK=$1 MS=$2 MODE=$3 NBUF=12 ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$1 $3 /" | tee -a $O/ab.txt; }
run 18944,3584 9,16,32 pc
run 18944,3584 16 g128
run 22016,4096 9,16,32 pc
run 20480,4096 16 pc
run 16384,4096 16,32 pc
run 16384,8192 16 pc
