#!/bin/bash
# HBM/fabric traffic of the two roofline kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md HBM section), written to gpurun_out/pmc_fetch, gpurun_out/pmc_write; summarised by
# tools/pmc_traffic.py into profiles/hbm_traffic.json.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python $R/tools/prof_calls.py --ms 1,16,4096 --iters 4 > $R/gpurun_out/pmc_$c.log 2>&1
done
