#!/bin/bash
# round 5, visit 29: the BASELINE layer at 64 / 128 / 256 tokens, the panel kernel's new grid order (default) against the plain one (tune.fused = 32), variants repeated
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v29; mkdir -p gpurun_out/r5v29; export TMPDIR=/tmp
T="[None, dict(fused=32), None, dict(fused=32), None, dict(fused=32)]"
NBUF=0 ROUNDS=8 ITERS=4 MS=64,128,256 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 NBUF=0 ROUNDS=8 ITERS=4 MS=64,128,256 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
