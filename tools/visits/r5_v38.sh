#!/bin/bash
# round 5, visit 38: the decode regime, cold, against fp16 on the same GPU (tools/decode_table.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v38; mkdir -p gpurun_out/r5v38; export TMPDIR=/tmp
timeout 800 python tools/decode_table.py 2>&1 | grep -v amdgpu.ids | tee $O/decode_table.txt
