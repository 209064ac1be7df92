#!/bin/bash
# round 4, visit 39: per-group decode (column kernel, re-quantiser in the dependent chain): prefetch depth, K split, at 1 / 8 / 16 tokens, BASELINE layer + two Llama layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v39; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=3,pf=2), dict(kernel=3,pf=3), dict(kernel=3,pf=4), dict(kernel=3,pf=6), dict(kernel=3,pf=8), dict(kernel=3,pf=12), dict(kernel=3,ksplit=2), dict(kernel=3,ksplit=2,pf=6), dict(kernel=3,ksplit=3), dict(kernel=1), dict(kernel=1,ksplit=8)]"
MODE=g128 MS=1,8,16 NBUF=5 ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
MODE=pc MS=1,8 NBUF=5 ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
NK=4096,4096 MODE=g128 MS=1,16 NBUF=12 ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 g128 /" | tee -a $O/ab.txt
NK=4096,11008 MODE=g128 MS=1,16 NBUF=12 ROUNDS=8 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x11008 g128 /" | tee -a $O/ab.txt
