#!/bin/bash
# round 5, visit 41: per-group, 9 ... 32 tokens: the stream kernel with 8 / 16 waves per workgroup (4 is its plan for one 16-token tile) and the column kernel with 16, cold
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v41; mkdir -p gpurun_out/r5v41; export TMPDIR=/tmp
T="[None, dict(kernel=1), dict(kernel=1,waves=8), dict(kernel=1,waves=16), dict(kernel=3), dict(kernel=3,waves=16), dict(kernel=1), dict(kernel=1,waves=8), dict(kernel=1,waves=16)]"
MODE=g128 NBUF=0 ROUNDS=8 ITERS=4 MS=12,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
for nk in 4096,11008 8192,8192 11008,4096; do
MODE=g128 NK=$nk NBUF=0 ROUNDS=8 ITERS=4 MS=16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$nk g128 /" | tee -a $O/ab.txt
done
T="[None, dict(kernel=1), dict(kernel=1,waves=16), dict(kernel=3,mt=2), dict(kernel=1), dict(kernel=1,waves=16)]"
MODE=g128 NBUF=0 ROUNDS=8 ITERS=4 MS=24,32 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
MODE=pc NBUF=0 ROUNDS=8 ITERS=4 MS=16 TUNES="[None, dict(kernel=1,waves=8), dict(kernel=1,waves=16), None, dict(kernel=1,waves=8), dict(kernel=1,waves=16)]" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
