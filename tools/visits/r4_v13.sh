#!/bin/bash
# round 4, visit 13: the tile walk's differential fuzz (tools/fuzz_walk.py; new GPU test) and a longer soak of both fuzzers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v13; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --durations=5 2>&1 | grep -v amdgpu.ids | tee $O/pytest_fuzz.log | tail -n 12
for s in 31 32; do SEED=$s SECONDS=100 timeout 400 python tools/fuzz_walk.py 2>&1 | grep -v amdgpu.ids | tee -a $O/fuzz_walk_soak.txt | tail -n 3; done
for s in 41 42; do SEED=$s SECONDS=100 timeout 400 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tee -a $O/fuzz_families_soak.txt | tail -n 3; done
