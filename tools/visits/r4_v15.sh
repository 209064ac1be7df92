#!/bin/bash
# round 4, visit 15: is the tile walk's gain capped by the power limit?  The same A/B (walk vs one tile per workgroup) on all-zero
# operands (no toggling in the matrix pipe: the clock stays up) and on the bench's random operands, 8192 and 32768 tokens
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v15; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[dict(kernel=5,glds=1), dict(kernel=5,glds=2)]"
for z in 0 1; do for nk in 4096,4096 11008,4096; do
  ZERO=$z NK=$nk NBUF=8 MS=8192,32768 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ZERO=$z NK=$nk pc   /" | tee -a $O/ab.txt
  ZERO=$z MODE=g128 NK=$nk NBUF=8 MS=8192 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ZERO=$z NK=$nk g128 /" | tee -a $O/ab.txt
done; done
ZERO=1 NBUF=4 MS=4096 ROUNDS=8 TUNES="[None]" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ZERO=1 BASELINE pc   /" | tee -a $O/ab.txt
ZERO=0 NBUF=4 MS=4096 ROUNDS=8 TUNES="[None]" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ZERO=0 BASELINE pc   /" | tee -a $O/ab.txt
