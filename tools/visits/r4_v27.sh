#!/bin/bash
# round 4, visit 27: dispatch check on Qwen2-7B (3584 / 18944 / kv 512) and Mistral-7B (14336) layer shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v27; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
WIDE_SHAPES=1 SHAPES=3584x3584,18944x3584,3584x18944,512x3584,14336x4096,4096x14336 MS=1,8,16,32,64,128,256,512,1024,2048,4096,8192 ITERS=9 timeout 2600 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_qwen_mistral.txt
grep -c "<--" $O/dispatch_check_qwen_mistral.txt; grep "<--" $O/dispatch_check_qwen_mistral.txt | sed 's/tiled *[0-9.na]* //' | cut -c1-330
