#!/bin/bash
# round 4, visit 18: the refitted 33 ... 64-token choice (panel vs stream) on ten layer shapes, both modes: automatic dispatch against every forced family
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v18; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
SHAPES=8192x21760,4096x4096,11008x4096,4096x11008,5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=40,48,56,64 ITERS=12 ROUNDS=4 timeout 1500 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_m64.txt
grep -c "<--" $O/dispatch_check_m64.txt; cut -c1-150 $O/dispatch_check_m64.txt | sed 's/tiled *[0-9.na]* //'
