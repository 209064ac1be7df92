#!/bin/bash
# round 4, visit 7: the split-K stress tests over the four hand-off variants (as shipped / acquire fence / release publish / both),
# then what each formal end costs at the points that use an in-launch split (interleaved A/B)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v07; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "stress or three_workgroups" 2>&1 | tail -5 | tee $O/pytest.txt
T="[dict(kernel=4), dict(kernel=4,fused=5), dict(kernel=4,fused=9), dict(kernel=4,fused=13)]"
NBUF=5 MS=128,256 ROUNDS=8 ITERS=8 TUNES="$T" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_handoff.txt
MODE=g128 NBUF=5 MS=128 ROUNDS=8 ITERS=8 TUNES="$T" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_handoff.txt
T="[dict(kernel=5,ksplit=2), dict(kernel=5,ksplit=2,fused=5), dict(kernel=5,ksplit=2,fused=9), dict(kernel=5,ksplit=2,fused=13)]"
MODE=g128 NBUF=5 MS=768,1024 ROUNDS=6 TUNES="$T" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_handoff.txt
T="[dict(kernel=2,bm=128,ksplit=2), dict(kernel=2,bm=128,ksplit=2,fused=5), dict(kernel=2,bm=128,ksplit=2,fused=9)]"
NBUF=5 MS=512 ROUNDS=6 TUNES="$T" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_handoff.txt
