#!/bin/bash
# round 4, visit 19: 9 ... 32 tokens (column kernel vs stream kernel + reduce) on ten layer shapes, both modes; BASELINE M=16 again with 5 and 8 rotating buffers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v19; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
SHAPES=8192x21760,4096x4096,11008x4096,4096x11008,5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=9,12,16,24,32 ITERS=12 ROUNDS=4 timeout 1500 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_m16.txt
grep -c "<--" $O/dispatch_check_m16.txt; cut -c1-100 $O/dispatch_check_m16.txt
T="[None, dict(kernel=3), dict(kernel=1)]"
for nb in 5 8; do NBUF=$nb MS=16,32 ROUNDS=8 TUNES="$T" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NBUF=$nb pc   /" | tee -a $O/ab_m16.txt; done
NBUF=8 MODE=g128 MS=16,32 ROUNDS=8 TUNES="$T" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NBUF=8 g128 /" | tee -a $O/ab_m16.txt
