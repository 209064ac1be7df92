#!/bin/bash
# round 4, visit 26: dispatch check on the merged projections of SURVEY 8 f-4 (q/k/v -> N = 12288, gate/up -> N = 22016, K = 4096) and two narrow layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v26; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
WIDE_SHAPES=1 SHAPES=12288x4096,22016x4096,2048x8192,1024x4096 MS=1,8,16,32,64,128,256,512,1024,2048,4096,8192 ITERS=9 timeout 2400 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_merged.txt
grep -c "<--" $O/dispatch_check_merged.txt; grep "<--" $O/dispatch_check_merged.txt | sed 's/tiled *[0-9.na]* //' | cut -c1-330
