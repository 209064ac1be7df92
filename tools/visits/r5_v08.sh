#!/bin/bash
# round 5, visit 8: the whole GPU suite + smoke + the driver's bench command on the library with the generated panel table; then the bench under rocprofv3 (kernel trace + stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v08; mkdir -p $O; rm -rf $O/*; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; wc -c $O/bench_driver.json; cp gpurun_out/bench_detail_n1.json $O/bench_detail_driver.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cp gpurun_out/bench_detail_n1.json $O/bench_detail_default.json
bash tools/profile_bench.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
