#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command + the per-(kernel, geometry) table + PMC pictures of the
# roofline kernels; everything under gpurun_out/prof_bench/, summaries are copied to profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/prof_bench; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# (a) the sweep alone (--no-llama): every (kernel, grid) row is ONE sweep point -- the table roofline.achieved is checked against;
# (b) the whole default command (with the Llama-2-7B block, whose layers share kernels and grids with the sweep's)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu --no-fp16 --no-pmc --no-llama --steps 40 --warmup 5 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python $R/tools/kernel_trace_table.py $O/trace > $O/bench_kernels_by_shape.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_llama -o bench -- python $R/bench.py --no-cpu --no-fp16 --no-pmc --steps 40 --warmup 5 > $O/bench_with_llama_under_rocprof.json 2> $O/bench_with_llama_under_rocprof.err
python $R/tools/kernel_trace_table.py $O/trace_llama > $O/bench_with_llama_kernels_by_shape.csv
cat $O/bench_kernels_by_shape.csv | cut -c1-220
cd $R
bash tools/pmc_kernel.sh wide_m4096 4096 pc '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh wide_m1024 1024 pc '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh panel_m128 128 pc '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh stream_m16 16 pc '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh column_m1 1 pc '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh wide_m4096_g128 4096 g128 '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh wide_m1024_g128 1024 g128 '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh wide_m4096_g128x 4096 g128x '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh wide_m1024_g128x 1024 g128x '{}' > /dev/null 2>&1
for t in wide_m4096 wide_m1024 panel_m128 stream_m16 column_m1 wide_m4096_g128 wide_m1024_g128 wide_m4096_g128x wide_m1024_g128x; do echo "#### $t"; cat gpurun_out/pmc_$t/summary.txt; done
