#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b13; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/check_variant.py --ms 4096,1000 --modes pc --tunes "[dict(kernel=5)]" --ref "dict(kernel=2)" > $O/check.log 2>&1; cat $O/check.log | grep -v amdgpu
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_sm0.so MS=2048,4096,8192 MODE=pc ROUNDS=6 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_sm0.so NK=11008,4096 MS=8192 MODE=pc ROUNDS=4 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
