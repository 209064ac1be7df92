#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b14; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/check_variant.py --ms 4096,1000 --tunes "[dict(kernel=5), dict(kernel=5, ksplit=2)]" --ref "dict(kernel=2)" > $O/check.log 2>&1; grep -v amdgpu $O/check.log
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so MS=2048,4096 MODE=pc ROUNDS=6 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_sm3.so,qqq_amd/libqqq_amd_prev.so MS=2048,4096 MODE=g128 ROUNDS=6 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
