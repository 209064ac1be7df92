#!/bin/bash
# round 5, visit 40: validation after the 16-wave column kernel for per-group decode: a short fuzz (the new variants are in tests/gpu_util.variants), smoke, the GPU suite, the driver's bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v40; mkdir -p gpurun_out/r5v40; export TMPDIR=/tmp
SEED=701 SECONDS=60 timeout 200 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/fuzz.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $O/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; cp gpurun_out/bench_detail_n1.json $O/bench_a_detail.json
tail -c 200 $O/bench_a.json
