#!/bin/bash
# round 5, visit 6: dispatch check with the 64-token m-block column (panel64) -- is the new candidate of the panel model chosen where it wins and only there?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v06; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
SHAPES=4096x4096,11008x4096,4096x11008,8192x8192,5120x5120,13824x5120,5120x13824,8192x21760,3584x18944,14336x4096,4096x14336,7168x7168,12288x4096,8192x3072 MS=80,96,128,160,192,256,320,384,512 ROUNDS=3 ITERS=9 timeout 1500 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids | tee $O/dispatch_check.txt
