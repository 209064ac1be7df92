#!/bin/bash
# round 3, box visit 2: in-kernel timeline of the wide kernel; pinned BASELINE sweep on reference-made operands
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b2; mkdir -p $O; export TMPDIR=/tmp
T=$PWD/qqq_amd/libqqq_amd_trace.so
QQQ_AMD_LIB=$T MS=4096,2048 timeout 300 python tools/trace_wide.py > $O/trace_pc.txt 2>&1; cat $O/trace_pc.txt
QQQ_AMD_LIB=$T MS=4096 TUNE="dict(kernel=5,pf=3)" timeout 300 python tools/trace_wide.py > $O/trace_pc_pf3.txt 2>&1; cat $O/trace_pc_pf3.txt
QQQ_AMD_LIB=$T MS=4096 MODE=g128 timeout 300 python tools/trace_wide.py > $O/trace_g128.txt 2>&1; cat $O/trace_g128.txt
QQQ_AMD_LIB=$T MS=4096 NK=11008,4096 timeout 300 python tools/trace_wide.py > $O/trace_llama.txt 2>&1; cat $O/trace_llama.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "pinned" > $O/pinned.log 2>&1; echo "rc=$?" >> $O/pinned.log; tail -8 $O/pinned.log
