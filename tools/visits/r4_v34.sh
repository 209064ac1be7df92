#!/bin/bash
# round 4, visit 34: EXPERIMENTAL parallel fold of the panel kernel's K slices (tune.fused bit 32): bit-exact? how much does it save at 64 ... 256 tokens?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v34; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 300 python tools/check_variant.py --ms 65,100,128 --tunes "[dict(kernel=4,mt=8,ksplit=4,fused=32), dict(kernel=4,mt=8,ksplit=2,fused=32)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
timeout 300 python tools/check_variant.py --nk 4096,4096 --ms 128,200,256 --tunes "[dict(kernel=4,mt=8,ksplit=4,fused=32), dict(kernel=4,mt=8,ksplit=2,fused=32)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
T="[None, dict(kernel=4,mt=8,ksplit=4), dict(kernel=4,mt=8,ksplit=4,fused=32), dict(kernel=4,mt=8,ksplit=2), dict(kernel=4,mt=8,ksplit=2,fused=32)]"
MS=96,128 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=128 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
NK=4096,4096 MS=128,256 NBUF=12 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc  /" | tee -a $O/ab.txt
NK=8192,8192 MS=128,256 NBUF=12 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/8192x8192 pc  /" | tee -a $O/ab.txt
