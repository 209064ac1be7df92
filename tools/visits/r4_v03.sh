#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v03; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/diag_chain.py --nk 4096,4096 --m 8192 --mode pc 2>&1 | grep -v amdgpu.ids | tee $O/diag_pc.txt
timeout 300 python tools/diag_chain.py --nk 4096,4096 --m 8192 --mode g128 --tune "dict(kernel=5,glds=2,mt=8)" --rows 128 2>&1 | grep -v amdgpu.ids | tee $O/diag_g128_mt8.txt
timeout 300 python tools/diag_chain.py --nk 4096,4096 --m 8192 --mode pc --tune "dict(kernel=5,glds=2,bm=128)" --cols 128 2>&1 | grep -v amdgpu.ids | tee $O/diag_pc_bm128.txt
