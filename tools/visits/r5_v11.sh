#!/bin/bash
# round 5, visit 11: EARLY FINISH of the panel kernel's uneven split (the longer last slice that finds every other deposit complete skips its ticket and the wait):
# bit-exactness, the panel stress tests, then A/B against the same library with the early finish switched off (tune.fused | 32) and with even slices (skew -1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v11; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
V="[dict(kernel=4), dict(kernel=4,skew=1), dict(kernel=4,skew=9), dict(kernel=4,ksplit=2,skew=20), dict(kernel=4,ksplit=3,skew=4), dict(kernel=4,mt=4,ksplit=4), dict(kernel=4,bm=256,ksplit=2,skew=3), dict(kernel=4,fused=32), dict()]"
timeout 200 python tools/check_variant.py --ms 40,64,65,100,128,200,300 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
timeout 200 python tools/check_variant.py --nk 4096,4160 --modes pc --ms 64,128,129 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
timeout 200 python tools/check_variant.py --nk 4096,4096 --ms 128,256 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
T="[None, dict(fused=32), dict(skew=-1), dict(kernel=4,skew=3), dict(kernel=4,skew=5), dict(kernel=4,skew=6), dict(kernel=4,skew=8)]"
MS=64,128,256 NBUF=5 ROUNDS=10 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=128 NBUF=5 ROUNDS=10 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
NK=4096,4096 MS=128,256 NBUF=12 ROUNDS=10 TUNES="[None, dict(fused=32), dict(skew=-1), dict(kernel=4,mt=8), dict(kernel=4,mt=8,fused=32), dict(kernel=4,mt=8,skew=-1)]" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc  /" | tee -a $O/ab.txt
NK=8192,8192 MS=128 NBUF=12 ROUNDS=10 TUNES="[None, dict(fused=32), dict(skew=-1)]" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/8192x8192 pc  /" | tee -a $O/ab.txt
