#!/bin/bash
# round 4, visit 8: dispatch check (interleaved) on the six extra layer shapes + the BASELINE / Llama shapes; the bench under
# rocprofv3 (kernel table) + PMC pictures of the sweep's kernels; PMC of the tile walk against the plain grid on a 4096-deep layer
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v08; mkdir -p $O; rm -rf $O/*; export TMPDIR=/tmp
WIDE_SHAPES=1 SHAPES=5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=1,16,64,128,512,2048,8192 ITERS=9 timeout 1500 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_shapes.txt; grep -c "<--" $O/dispatch_check_shapes.txt
WIDE_SHAPES=1 SHAPES=8192x21760,4096x4096,11008x4096,4096x11008 MS=1,16,64,128,256,512,1024,2048,4096,8192 ITERS=9 timeout 1500 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check.txt; grep -c "<--" $O/dispatch_check.txt
timeout 2400 bash tools/profile_bench.sh > $O/profile_bench.txt 2>&1; cp -r gpurun_out/prof_bench/bench_kernels_by_shape.csv gpurun_out/prof_bench/bench_under_rocprof.json $O/ 2>/dev/null; cp gpurun_out/prof_bench/trace/*stats*.csv $O/ 2>/dev/null
for t in wide_m4096 wide_m1024 panel_m128 stream_m16 column_m1 wide_m4096_g128 wide_m1024_g128; do cp gpurun_out/pmc_$t/summary.txt $O/pmc_$t.txt; done
bash tools/pmc_kernel.sh walk_4096x4096_m8192 8192 pc '{"kernel":5,"glds":2}' 4096,4096 > /dev/null 2>&1; cp gpurun_out/pmc_walk_4096x4096_m8192/summary.txt $O/pmc_walk_4096x4096_m8192.txt
bash tools/pmc_kernel.sh plain_4096x4096_m8192 8192 pc '{"kernel":5,"glds":1}' 4096,4096 > /dev/null 2>&1; cp gpurun_out/pmc_plain_4096x4096_m8192/summary.txt $O/pmc_plain_4096x4096_m8192.txt
ls $O
