#!/bin/bash
# round 3: bench line, then the bench under rocprofv3 (per-kernel table) and the PMC pictures of every sweep point's kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
bash tools/visits/r3_bench.sh ${1:-r3final}
timeout 3000 bash tools/profile_bench.sh > gpurun_out/${1:-r3final}/profile_bench.txt 2>&1; tail -150 gpurun_out/${1:-r3final}/profile_bench.txt
