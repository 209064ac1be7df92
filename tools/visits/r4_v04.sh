#!/bin/bash
# round 4, visit 4: seam timeline of the persistent tile walk against the plain kernel's phases (trace build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v04; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
export QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_trace.so
for nk in 4096,4096 4096,11008; do
NK=$nk MS=8192 timeout 300 python tools/trace_chain.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_chain.txt
NK=$nk MS=8192 TUNE="dict(kernel=5,glds=1)" timeout 300 python tools/trace_wide.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_plain.txt
done
NK=4096,4096 MS=32768 timeout 300 python tools/trace_chain.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_chain.txt
NK=8192,21760 MS=4096 timeout 300 python tools/trace_chain.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_chain.txt
NK=8192,21760 MS=4096 TUNE="dict(kernel=5,glds=1)" timeout 300 python tools/trace_wide.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_plain.txt
