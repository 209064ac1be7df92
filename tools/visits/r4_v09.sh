#!/bin/bash
# round 4, visit 9: XCD-local split-K deposits of the wide kernel (slices that find each other on one XCD keep their partial tile in
# that XCD's L2): stress tests, bit-exactness, and the hand-off's cost against forced write-through (tune.fused bit 4 = 16)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v09; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wide_inlaunch or wide_every" 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python tools/check_variant.py --ms 700,1024,2000 --tunes "[dict(kernel=5,ksplit=2), dict(kernel=5,ksplit=3), dict(kernel=5,ksplit=2,bm=128), dict(kernel=5,ksplit=2,fused=17)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
T="[dict(kernel=5), dict(kernel=5,mt=8), dict(kernel=5,ksplit=2), dict(kernel=5,ksplit=2,fused=17), dict(kernel=5,bm=128,ksplit=2), dict(kernel=5,bm=128,ksplit=2,fused=17)]"
NBUF=5 MS=512,768,1024,1536 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_local.txt
MODE=g128 NBUF=5 MS=512,768,1024,1536 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_local.txt
