#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/pmc_kernel.sh panel256_m1024 1024 pc '{"kernel":4,"mt":8,"bm":256,"ksplit":1,"pf":3}' > /dev/null 2>&1
bash tools/pmc_kernel.sh tiled_m4096 4096 pc '{"kernel":2}' > /dev/null 2>&1
bash tools/pmc_kernel.sh tiled_m1024 1024 pc '{"kernel":2}' > /dev/null 2>&1
bash tools/pmc_kernel.sh panel_m128 128 pc '{"kernel":4}' > /dev/null 2>&1
for t in panel256_m1024 tiled_m4096 tiled_m1024 panel_m128; do cat gpurun_out/pmc_$t/summary.txt | grep -v "^   [A-Z]"; done
