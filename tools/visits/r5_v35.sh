#!/bin/bash
# round 5, visit 35: last validation -- smoke, the GPU suite, the driver's bench command (the library is visit 34's, rebuilt after a comment-only change)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v35; mkdir -p gpurun_out/r5v35; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $O/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err
tail -c 200 $O/bench_a.json
