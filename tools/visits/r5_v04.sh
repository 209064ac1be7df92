#!/bin/bash
# round 5, visit 4: the stream kernel's in-launch split-K through arrival-order slots (fused=3: the last arrival keeps its tile in LDS, uneven slices by `skew` in 64-k steps)
# against slabs + reduce launch (the automatic choice so far) at 9 ... 64 tokens; bit-exactness first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v04; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
V="[dict(kernel=1,fused=3), dict(kernel=1,fused=3,skew=-1), dict(kernel=1,fused=3,skew=12), dict(kernel=1,ksplit=7,fused=3,skew=3), dict(kernel=1,ksplit=2,waves=4,fused=3,skew=40), dict(kernel=1,mt=2,ksplit=3,fused=3), dict(kernel=1,fused=1)]"
timeout 300 python tools/check_variant.py --ms 1,16,33,64,100 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
timeout 300 python tools/check_variant.py --nk 4096,4160 --modes pc --ms 16,40 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
timeout 300 python tools/check_variant.py --nk 11008,4096 --ms 9,64 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
T="[None, dict(kernel=1), dict(kernel=1,fused=3,skew=-1), dict(kernel=1,fused=3,skew=4), dict(kernel=1,fused=3,skew=7), dict(kernel=1,fused=3,skew=10), dict(kernel=1,fused=3,skew=14), dict(kernel=1,fused=3,skew=20), dict(kernel=3)]"
MS=9,16,32,48,64 NBUF=5 ROUNDS=10 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=16,32,64 NBUF=5 ROUNDS=10 TUNES="$T" timeout 500 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
T="[None, dict(kernel=1), dict(kernel=1,fused=3,skew=-1), dict(kernel=1,fused=3), dict(kernel=1,fused=3,skew=4), dict(kernel=1,fused=3,skew=8), dict(kernel=1,fused=3,skew=16)]"
for NK in 4096,11008 4096,4096 11008,4096 8192,8192 5120,13824; do
NK=$NK MS=16,32,64 NBUF=12 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$NK pc   /" | tee -a $O/ab_shapes.txt
done
NK=4096,11008 MODE=g128 MS=16,64 NBUF=12 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096,11008 g128 /" | tee -a $O/ab_shapes.txt
