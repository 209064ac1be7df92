#!/bin/bash
# round 4, visit 29: dispatch check on Yi-34B (7168 / 20480), Phi-3 (3072 / 8192), Llama-70B k / v (N = 1024, K = 8192) and Qwen2-72B up (29568 x 8192) shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v29; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
WIDE_SHAPES=1 SHAPES=7168x7168,20480x7168,7168x20480,3072x3072,8192x3072,3072x8192,1024x8192,29568x8192 MS=1,8,16,32,64,128,256,512,1024,4096 ITERS=9 timeout 2800 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_more_models.txt
grep -c "<--" $O/dispatch_check_more_models.txt; grep "<--" $O/dispatch_check_more_models.txt | sed 's/tiled *[0-9.na]* //' | cut -c1-330
