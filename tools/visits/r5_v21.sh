#!/bin/bash
# round 5, visit 21: the library with non-temporal weight loads in the column kernel and the panel kernel's 64-token m-blocks (default build) against a build without
# (-DQQQ_W_NT=0 -> qqq_amd/libabl_plain.so); the column kernel's prefetch depth once more; then EVERY dispatch grid again on this library (the data the cost models are held against)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v21; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
L=qqq_amd/libabl_plain.so,qqq_amd/libqqq_amd.so,qqq_amd/libabl_plain.so,qqq_amd/libqqq_amd.so
NBUF=5 LIBS=$L ROUNDS=8 ITERS=4 MS=1,4,8,64 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 NBUF=5 LIBS=$L ROUNDS=8 ITERS=4 MS=1,8,16,32 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
NK=4096,4096 NBUF=24 LIBS=$L ROUNDS=8 ITERS=4 MS=1,16,64,128,256 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x4096 pc /" | tee -a $O/ab.txt
NK=4096,11008 NBUF=12 LIBS=$L ROUNDS=8 ITERS=4 MS=1,16,64,128 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096x11008 pc /" | tee -a $O/ab.txt
NK=11008,4096 NBUF=12 LIBS=$L ROUNDS=8 ITERS=4 MS=1,16,64,128 timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/11008x4096 pc /" | tee -a $O/ab.txt
T="[None, dict(kernel=3,pf=2), dict(kernel=3,pf=4), dict(kernel=3,pf=5), dict(kernel=3,pf=6), dict(kernel=3,pf=8), dict(kernel=3,pf=12), dict(kernel=1), dict(kernel=1,ksplit=4)]"
NBUF=5 ROUNDS=8 ITERS=4 MS=1,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab_column.txt
MODE=g128 NBUF=5 ROUNDS=8 ITERS=4 MS=1,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab_column.txt
run() { name=$1; shift; env "$@" ROUNDS=3 ITERS=9 timeout 900 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_check_$name.txt; wc -l $O/dispatch_check_$name.txt; }
run main SHAPES=8192x21760,4096x4096,11008x4096,4096x11008 MS=1,16,64,128,256,512,1024,2048,4096,8192 WIDE_SHAPES=1
run shapes SHAPES=5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=1,16,64,128,512,2048,8192 WIDE_SHAPES=1
run m16 SHAPES=8192x21760,4096x4096,11008x4096,4096x11008,5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=9,12,16,24,32 WIDE_SHAPES=0
run m64 SHAPES=8192x21760,4096x4096,11008x4096,4096x11008,5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=40,48,56,64 WIDE_SHAPES=0
run merged SHAPES=12288x4096,22016x4096,2048x8192,1024x4096 MS=1,8,16,32,64,128,256,512,1024,2048,4096,8192 WIDE_SHAPES=1
run mid SHAPES=8192x21760 MS=320,512,640,768,1024,1280,1536,2048,3072 WIDE_SHAPES=1
run mid_shapes SHAPES=4096x4096,11008x4096,4096x11008,5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=96,192,256,384,768,1024,1536,4096 WIDE_SHAPES=1
run more_models SHAPES=7168x7168,20480x7168,7168x20480,3072x3072,8192x3072,3072x8192,1024x8192,29568x8192 MS=1,8,16,32,64,128,256,512,1024,4096 WIDE_SHAPES=1
run qwen_mistral SHAPES=3584x3584,18944x3584,3584x18944,512x3584,14336x4096,4096x14336 MS=1,8,16,32,64,128,256,512,1024,2048,4096,8192 WIDE_SHAPES=1
run panel64 SHAPES=4096x4096,11008x4096,4096x11008,8192x8192,5120x5120,13824x5120,5120x13824,8192x21760,3584x18944,14336x4096,4096x14336,7168x7168,12288x4096,8192x3072 MS=80,96,128,160,192,256,320,384,512
