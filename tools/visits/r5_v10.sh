#!/bin/bash
# round 5, visit 10: in-kernel phase timeline of the panel kernel at 128 tokens (trace build, -DQQQ_PANEL_TRACE) with even K slices (skew -1) and with the automatic skew
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v10; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
for T in "dict(skew=-1)" "None"; do
QQQ_AMD_LIB=qqq_amd/libqqq_amd_trace.so MS=128 TUNE="$T" timeout 200 python tools/trace_panel.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$T] /" | tee -a $O/timeline.txt
QQQ_AMD_LIB=qqq_amd/libqqq_amd_trace.so MS=128 MODE=g128 TUNE="$T" timeout 200 python tools/trace_panel.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$T] /" | tee -a $O/timeline.txt
QQQ_AMD_LIB=qqq_amd/libqqq_amd_trace.so MS=128 NK=4096,4096 TUNE="dict(kernel=4,mt=8,$( [ "$T" = None ] && echo skew=0 || echo skew=-1 ))" timeout 200 python tools/trace_panel.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[4096x4096 $T] /" | tee -a $O/timeline.txt
done
