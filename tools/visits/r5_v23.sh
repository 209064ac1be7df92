#!/bin/bash
# round 5, visit 23: nt weight loads measured WITHOUT help from the Infinity Cache: 12 weight copies (1.07 GB) per variant group, 640 MB of unrelated traffic in front of
# every group (FLUSH=1); and what nt does to Infinity Cache residency: 2 copies (178 MB, fits), flushed in front of every group of 8 calls
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v23; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
L=qqq_amd/libabl_plain.so,qqq_amd/libqqq_amd.so,qqq_amd/libabl_plain.so,qqq_amd/libqqq_amd.so
echo "## 2 weight copies, flushed in front of each group of 8 calls" | tee -a $O/ab.txt
FLUSH=1 NBUF=2 LIBS=$L ROUNDS=6 ITERS=8 MS=1 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
echo "## 12 weight copies, 12 calls per group, flushed in front of each group" | tee -a $O/ab.txt
FLUSH=1 NBUF=12 LIBS=$L ROUNDS=6 ITERS=12 MS=1,8,16,64 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 FLUSH=1 NBUF=12 LIBS=$L ROUNDS=6 ITERS=12 MS=1,16 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
NK=4096,11008 FLUSH=1 NBUF=24 LIBS=$L ROUNDS=6 ITERS=24 MS=1,16 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x11008 pc /" | tee -a $O/ab.txt
NK=4096,4096 FLUSH=1 NBUF=48 LIBS=$L ROUNDS=6 ITERS=48 MS=1,16,64,128 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc /" | tee -a $O/ab.txt
echo "## column kernel vs stream kernel, prefetch depths (final library), 12 copies, flushed" | tee -a $O/ab.txt
T="[None, dict(kernel=3,pf=2), dict(kernel=3,pf=3), dict(kernel=3,pf=4), dict(kernel=1), dict(kernel=3,pf=2), dict(kernel=3,pf=3), dict(kernel=3,pf=4), dict(kernel=1)]"
FLUSH=1 NBUF=12 ROUNDS=6 ITERS=12 MS=1,8,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 FLUSH=1 NBUF=12 ROUNDS=6 ITERS=12 MS=1,8,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
