#!/bin/bash
# round 3, box visit 3: whole -m gpu suite with the wide kernel in the automatic dispatch, epilogue timeline, PMC pictures, dispatch check
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b3; mkdir -p $O; export TMPDIR=/tmp
T=$PWD/qqq_amd/libqqq_amd_trace.so
QQQ_AMD_LIB=$T MS=4096 timeout 300 python tools/trace_wide.py > $O/trace_pc.txt 2>&1; grep -E "==|prologue|main loop|epilogue|kernel end" $O/trace_pc.txt
QQQ_AMD_LIB=$T MS=4096 NK=11008,4096 timeout 300 python tools/trace_wide.py > $O/trace_llama.txt 2>&1; grep -E "==|prologue|main loop|epilogue|kernel end" $O/trace_llama.txt
SHAPES=8192x21760 MS=768,1024,1280,1536,2048,3072,4096,8192 ITERS=8 timeout 900 python tools/dispatch_check.py > $O/dispatch_base.txt 2>&1; cat $O/dispatch_base.txt
SHAPES=4096x4096,11008x4096,4096x11008 MS=1024,2048,4096,8192,32768 ITERS=6 timeout 900 python tools/dispatch_check.py > $O/dispatch_llama.txt 2>&1; cat $O/dispatch_llama.txt
timeout 1500 bash tools/pmc_kernel.sh wide_m4096 4096 pc '{"kernel":5}' > $O/pmc_wide_m4096.txt 2>&1; tail -40 $O/pmc_wide_m4096.txt
timeout 1500 bash tools/pmc_kernel.sh wide_m4096_g128 4096 g128 '{"kernel":5}' > $O/pmc_wide_m4096_g128.txt 2>&1; tail -40 $O/pmc_wide_m4096_g128.txt
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -n 16 $O/pytest_gpu.log
