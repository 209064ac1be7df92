#!/bin/bash
# round 5, visit 12: last validation of the final library on one more box -- smoke, the GPU suite, the driver's bench command twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v12; mkdir -p $O; rm -rf $O/*; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.json 2> $O/bench_$i.err; wc -c $O/bench_$i.json; done
