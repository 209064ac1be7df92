#!/bin/bash
# round 5, visit 2: (a) what bounds the 128-token loop -- ablation COMBINATIONS of the panel kernel (10 = no global loads in the loop, 21 = no barrier / unpack /
# fragment reads: memory path + MFMAs only, 17 = no barrier / fragment reads); (b) uneven K slices (tune.skew) at 64 ... 256 tokens; (c) 128-column strips as
# 4 waves x 64 columns (pw=2); (d) 16 tokens: column (single / paired steps, ring depths) against stream.  Bit-exactness of every new variant first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v02; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 300 python tools/check_variant.py --ms 65,100,128 --tunes "[dict(kernel=4,skew=3), dict(kernel=4,skew=9), dict(kernel=4,ksplit=2,skew=20), dict(kernel=4,bm=128,pw=2), dict(kernel=4,bm=128,pw=2,ksplit=2,skew=5)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
timeout 300 python tools/check_variant.py --ms 1,7,16,29 --tunes "[dict(kernel=3,stages=2), dict(kernel=3,stages=2,pf=4), dict(kernel=3,stages=2,pf=6,ksplit=3), dict(kernel=3,stages=2,mt=2)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
timeout 300 python tools/check_variant.py --nk 4096,4160 --modes pc --ms 16,128 --tunes "[dict(kernel=3,stages=2), dict(kernel=4,skew=2), dict(kernel=4,bm=128,pw=2,skew=1)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
# (a) ablation combinations, automatic dispatch (the 8-wave 32-column shape, 4 K slices)
L=qqq_amd/libqqq_amd.so,qqq_amd/libabl_10.so,qqq_amd/libabl_21.so,qqq_amd/libabl_17.so
NBUF=5 LIBS=$L ROUNDS=6 ITERS=4 MS=64,128 timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ABLATE pc   /" | tee -a $O/ablate.txt
# (b) + (c)
T="[None, dict(kernel=4,skew=2), dict(kernel=4,skew=3), dict(kernel=4,skew=4), dict(kernel=4,skew=6), dict(kernel=4,skew=9), dict(kernel=4,bm=128,pw=2), dict(kernel=4,bm=128,pw=2,skew=3), dict(kernel=4,bm=128,pw=2,ksplit=2), dict(kernel=4,ksplit=2,skew=12)]"
MS=64,128,256 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=64,128 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
NK=4096,4096 MS=128,256 NBUF=12 ROUNDS=8 TUNES="[None, dict(kernel=4,skew=1), dict(kernel=4,skew=2), dict(kernel=4,skew=3), dict(kernel=4,bm=128,pw=2)]" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc  /" | tee -a $O/ab.txt
# (d)
T="[None, dict(kernel=3), dict(kernel=3,stages=2), dict(kernel=3,pf=4), dict(kernel=3,stages=2,pf=4), dict(kernel=3,stages=2,pf=6), dict(kernel=3,pf=6), dict(kernel=1), dict(kernel=1,fused=3)]"
MS=8,16,24 NBUF=5 ROUNDS=10 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab16.txt
MODE=g128 MS=16 NBUF=5 ROUNDS=10 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab16.txt
NK=4096,11008 MS=16,24 NBUF=12 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x11008 pc /" | tee -a $O/ab16.txt
