#!/bin/bash
# round 4, visit 14: 128 ... 256 tokens as m-blocks of 64 rows, two workgroups per CU (panel kernel mt=4, 4 K slices) against the dispatch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v14; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=4,mt=4,ksplit=4), dict(kernel=4,mt=4,ksplit=2), dict(kernel=4,mt=4,ksplit=3), dict(kernel=4,mt=4,ksplit=4,pf=2), dict(kernel=4,mt=4,ksplit=4,stages=3,pf=3), dict(kernel=4,mt=4,ksplit=4,bm=256), dict(kernel=4,mt=2,ksplit=4), dict(kernel=4,mt=8,ksplit=4,stages=3,pf=3)]"
MS=128,192,256 ROUNDS=6 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_pc.txt
MODE=g128 MS=128,256 ROUNDS=6 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_g128.txt
