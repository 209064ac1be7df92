#!/bin/bash
# round 4, visit 43: the whole -m gpu suite three times in a row on one more box (flakiness check of the final library), then the two fuzzers with fresh seeds
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v43; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/suite.log; done
for s in 51 52; do SEED=$s SECONDS=60 timeout 300 python tools/fuzz_walk.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/fuzz.log; done
for s in 61 62; do SEED=$s SECONDS=60 timeout 300 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/fuzz.log; done
