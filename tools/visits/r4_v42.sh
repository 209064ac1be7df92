#!/bin/bash
# round 4, visit 42: strips per XCD panel of the wide kernel's tile order (pw) once more on the final library: time and fabric traffic go together?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v42; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[dict(kernel=5,pw=4), dict(kernel=5,pw=8), dict(kernel=5,pw=16), dict(kernel=5,pw=32)]"
MS=4096,8192 NBUF=5 ROUNDS=8 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=4096 NBUF=5 ROUNDS=8 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
NK=4096,4096 MS=8192,32768 NBUF=12 ROUNDS=8 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc  /" | tee -a $O/ab.txt
NK=11008,4096 MS=8192 NBUF=12 ROUNDS=8 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/11008x4096 pc /" | tee -a $O/ab.txt
NK=4096,11008 MS=8192 NBUF=12 ROUNDS=8 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x11008 pc /" | tee -a $O/ab.txt
