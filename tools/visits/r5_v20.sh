#!/bin/bash
# round 5, visit 20: non-temporal weight loads in every family (measurement build -DQQQ_W_NT=15 -> qqq_amd/libabl_nt.so) against the shipped library, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v20; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
L=qqq_amd/libqqq_amd.so,qqq_amd/libabl_nt.so,qqq_amd/libqqq_amd.so,qqq_amd/libabl_nt.so
NBUF=5 LIBS=$L ROUNDS=8 ITERS=4 MS=1,16,64,128 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
NBUF=5 LIBS=$L ROUNDS=6 ITERS=4 MS=1024,4096 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 NBUF=5 LIBS=$L ROUNDS=6 ITERS=4 MS=1,16,128,4096 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
NK=4096,4096 NBUF=24 LIBS=$L ROUNDS=8 ITERS=4 MS=1,16,128,1024 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc /" | tee -a $O/ab.txt
NK=4096,11008 NBUF=12 LIBS=$L ROUNDS=8 ITERS=4 MS=1,128,1024 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x11008 pc /" | tee -a $O/ab.txt
