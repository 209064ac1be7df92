#!/bin/bash
# round 4, visit 20: dispatch check at the token counts the earlier sets skipped (96, 192, 256, 384, 768, 1024, 1536, 4096) on nine layer shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v20; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
WIDE_SHAPES=1 SHAPES=4096x4096,11008x4096,4096x11008,5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=96,192,256,384,768,1024,1536,4096 ITERS=9 timeout 2400 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_mid_shapes.txt
grep -c "<--" $O/dispatch_check_mid_shapes.txt; grep "<--" $O/dispatch_check_mid_shapes.txt | cut -c1-60,330-420
