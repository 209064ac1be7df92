#!/bin/bash
# round 3, box visit 11: compiled torch binding -- whole GPU suite through it, host overhead per call
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b11; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 300 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; cat $O/host_overhead.txt
timeout 2400 python -m pytest tests -m gpu -q -x --durations=4 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -n 9 $O/pytest_gpu.log
QQQ_AMD_FORCE_DISPATCHER=1 timeout 1200 python -m pytest tests/test_gpu_qlinear.py tests/test_gpu_probe.py -m gpu -q -x > $O/pytest_disp.log 2>&1; echo "pytest(dispatcher) rc=$?" >> $O/pytest_disp.log; tail -n 3 $O/pytest_disp.log
