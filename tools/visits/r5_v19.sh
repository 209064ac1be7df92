#!/bin/bash
# round 5, visit 19: the L2 fill micro-benchmark with cache-policy bits on the loads (sc0 / sc1 / nt), + part-chip stream, scalar-path prefetch probes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v19; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/l2_fill_bench.hip -o /tmp/l2fb 2>/dev/null || exit 1
timeout 200 /tmp/l2fb 16 | tee $O/l2_fill.txt
timeout 200 /tmp/l2fb 16 > $O/l2_fill_again.txt
