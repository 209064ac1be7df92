#!/bin/bash
# round 5, visit 13: 256-column strips in 5 ... 8 K slices at 128 tokens (half the activation bytes per weight byte, twice the hand-off) -- what the existing serial fold makes of it
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v13; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=4,bm=256,ksplit=4), dict(kernel=4,bm=256,ksplit=6,skew=-1), dict(kernel=4,bm=256,ksplit=8,skew=-1), dict(kernel=4,bm=256,ksplit=8,skew=2), dict(kernel=4,bm=256,ksplit=8,skew=4), dict(kernel=4,bm=256,ksplit=8,skew=7), dict(kernel=4,bm=256,pw=2,ksplit=8,skew=-1), dict(kernel=4,bm=256,pw=2,ksplit=8,skew=4)]"
MS=128 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=128 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
timeout 200 python tools/check_variant.py --ms 128 --tunes "[dict(kernel=4,bm=256,ksplit=8,skew=4), dict(kernel=4,bm=256,pw=2,ksplit=8,skew=4)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
