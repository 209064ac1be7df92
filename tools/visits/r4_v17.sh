#!/bin/bash
# round 4, visit 17: dispatch check of the FINAL library (automatic choice against every forced family, interleaved) on both shape sets
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v17; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
WIDE_SHAPES=1 SHAPES=5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=1,16,64,128,512,2048,8192 ITERS=9 timeout 1500 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_shapes.txt; grep -c "<--" $O/dispatch_check_shapes.txt
WIDE_SHAPES=1 SHAPES=8192x21760,4096x4096,11008x4096,4096x11008 MS=1,16,64,128,256,512,1024,2048,4096,8192 ITERS=9 timeout 1500 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check.txt; grep -c "<--" $O/dispatch_check.txt
grep -h "<--" $O/*.txt | cut -c1-60,200-520
