#!/bin/bash
# round 5, visit 14: are the wide kernel's knobs (weight ring depth pf, tile-order panel width pw) still at their best values at the two large BASELINE points?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v14; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=5,pf=8), dict(kernel=5,pw=4), dict(kernel=5,pw=16), dict(kernel=5,pw=32), dict(kernel=5,glds=2), dict(kernel=5,pf=8,pw=16)]"
MS=4096 NBUF=5 ROUNDS=6 ITERS=4 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=4096 NBUF=5 ROUNDS=6 ITERS=4 TUNES="[None, dict(kernel=5,pf=4), dict(kernel=5,pw=4), dict(kernel=5,pw=16), dict(kernel=5,pw=32), dict(kernel=5,glds=2)]" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
T="[None, dict(kernel=5,bm=128,pf=8), dict(kernel=5,bm=128,pw=4), dict(kernel=5,bm=128,pw=16), dict(kernel=5,mt=8), dict(kernel=5,mt=8,pf=8), dict(kernel=5,ksplit=2), dict(kernel=5,ksplit=2,pf=8)]"
MS=1024 NBUF=5 ROUNDS=6 ITERS=4 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
