#!/bin/bash
# round 4, visit 10: panel kernel with the XCD-aware 1-D grid + XCD-local deposits: stress tests, bit-exactness, timing against the
# old grid / write-through (tune.fused bit 4 = 16) at the split points
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v10; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "stress or three_workgroups or golden or random_shapes or panel_two" 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python tools/check_variant.py --ms 33,64,100,128,200,256,300 --tunes "[dict(kernel=4), dict(kernel=4,ksplit=3), dict(kernel=4,bm=256,ksplit=2), dict(kernel=4,mt=4,ksplit=4), dict(kernel=4,fused=17)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
T="[dict(kernel=4), dict(kernel=4,fused=17), dict(kernel=1)]"
NBUF=5 MS=64,96,128,192,256 ROUNDS=8 ITERS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_panel.txt
MODE=g128 NBUF=5 MS=64,128,256 ROUNDS=8 ITERS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_panel.txt
NK=4096,4096 NBUF=12 MS=128,256,512 ROUNDS=8 ITERS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_panel.txt
