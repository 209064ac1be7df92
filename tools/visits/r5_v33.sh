#!/bin/bash
# round 5, visit 33: the bench under rocprofv3 (kernel trace + stats) and the PMC pictures once more, on the round's last library (1024 tokens now run as 256 x 256 tiles in two K slices)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v33; mkdir -p gpurun_out/r5v33; export TMPDIR=/tmp
bash tools/profile_bench.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
