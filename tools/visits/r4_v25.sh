#!/bin/bash
# round 4, visit 25: the bench under rocprofv3 (kernel table) + PMC pictures of the sweep's kernels on the FINAL library
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v25; mkdir -p $O; rm -rf $O/*; export TMPDIR=/tmp
timeout 2400 bash tools/profile_bench.sh > $O/profile_bench.txt 2>&1; cp -r gpurun_out/prof_bench/bench_kernels_by_shape.csv gpurun_out/prof_bench/bench_with_llama_kernels_by_shape.csv gpurun_out/prof_bench/bench_under_rocprof.json $O/ 2>/dev/null; cp gpurun_out/prof_bench/trace/*stats*.csv $O/ 2>/dev/null; for f in gpurun_out/prof_bench/trace_llama/*kernel_stats*.csv; do cp $f $O/bench_with_llama_kernel_stats.csv; done
for t in wide_m4096 wide_m1024 panel_m128 stream_m16 column_m1 wide_m4096_g128 wide_m1024_g128; do cp gpurun_out/pmc_$t/summary.txt $O/pmc_$t.txt; done
ls $O; grep -i "wide" $O/bench_kernels_by_shape.csv | cut -c1-160
