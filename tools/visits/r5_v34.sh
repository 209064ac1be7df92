#!/bin/bash
# round 5, visit 34: soak of the round's last library -- differential fuzz (four fresh seeds, 100 s each, idle and loaded chip), the tile-walk fuzzer, the GPU suite twice more
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v34; mkdir -p gpurun_out/r5v34; export TMPDIR=/tmp
for seed in 611 612 613 614; do SEED=$seed SECONDS=100 timeout 300 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/fuzz.txt; done
SEED=77 SECONDS=60 timeout 200 python tools/fuzz_walk.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/fuzz.txt
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee -a $O/suite.txt; done
