#!/bin/bash
# round 5, visit 15: weight ring depth of the wide kernel (pf 4 vs 8) at 4096 / 1024 tokens with the variants repeated in the list (order effects of the interleaved loop)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v15; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[dict(kernel=5,pf=4), dict(kernel=5,pf=8), dict(kernel=5,pf=4), dict(kernel=5,pf=8), None, dict(kernel=5,pf=8), dict(kernel=5,pf=4)]"
MS=4096 NBUF=5 ROUNDS=8 ITERS=4 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=4096 NBUF=5 ROUNDS=8 ITERS=4 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
T="[dict(kernel=5,bm=128,pf=4), dict(kernel=5,bm=128,pf=8), dict(kernel=5,bm=128,pf=4), dict(kernel=5,bm=128,pf=8), None]"
MS=1024 NBUF=5 ROUNDS=8 ITERS=4 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
NK=4096,4096 MS=8192 NBUF=12 ROUNDS=8 ITERS=4 TUNES="[dict(kernel=5,pf=4), dict(kernel=5,pf=8), dict(kernel=5,pf=4), dict(kernel=5,pf=8)]" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc /" | tee -a $O/ab.txt
