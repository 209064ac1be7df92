#!/bin/bash
# One GPU-box visit: probes, parity, tuning sweep, bench, rocprof.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > gpurun_out/box.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 300 python -m pytest tests/test_gpu_probe.py -m gpu -q -s > gpurun_out/pytest_probe.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s > gpurun_out/pytest_parity.log 2>&1
timeout 600 python -m pytest tests/test_gpu_qlinear.py -m gpu -q -s > gpurun_out/pytest_qlinear.log 2>&1
for f in smoke pytest_probe pytest_parity pytest_qlinear; do echo "== $f"; tail -n 4 gpurun_out/$f.log; done
if [ "${SKIP_TUNE:-0}" != "1" ]; then
  timeout 900 python tools/tune_sweep.py ${TUNE_ARGS:-} > gpurun_out/tune.txt 2>&1
  tail -3 gpurun_out/tune.txt
fi
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 600 gpurun_out/bench.err; head -c 3000 gpurun_out/bench.json
if [ "${SKIP_PROF:-0}" != "1" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu --no-fp16 --steps 10 --warmup 3 > "$OLDPWD/gpurun_out/prof_run.log" 2>&1
  cd "$OLDPWD"
  find gpurun_out/prof -name "*stats*" | head; 
  for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 "$f"; done
fi
