#!/bin/bash
# round 5, visit 28: validation of the final library (generated rates on the cold grids, XCD-gathered K slices, nt weight loads): differential fuzz (the panel kernel's new
# grid order meets random strip counts there), smoke, the GPU suite, the driver's bench command twice, the bench under rocprofv3 + PMC pictures
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v28; mkdir -p gpurun_out/r5v28; export TMPDIR=/tmp
for seed in 501 502 503; do SEED=$seed SECONDS=50 timeout 200 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/fuzz.txt; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; cp gpurun_out/bench_detail_n1.json $O/bench_a_detail.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b.json 2> $O/bench_b.err
tail -c 300 $O/bench_b.json
bash tools/profile_bench.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
