#!/bin/bash
# round 3, batch 23: 256 x 128 tiles of the wide kernel (32 columns per wave): parity, then timing against the 128-token tiles
out=gpurun_out/r3b23; mkdir -p $out
timeout 900 python tools/check_variant.py --ms 1024,1000,300,2100,4096 --tunes "[dict(kernel=5,bm=128), dict(kernel=5,bm=128,pf=8), dict(kernel=5,bm=128,ksplit=2,pw=4), dict(kernel=5,bm=128,ksplit=3)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | tee $out/check.log
timeout 300 python tools/check_variant.py --nk 320,640 --ms 300,70 --tunes "[dict(kernel=5,bm=128), dict(kernel=5,bm=128,ksplit=2)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | tee -a $out/check.log
grep -q MISMATCH $out/check.log && exit 1
NBUF=8 MS=512,640,768,1024,1152,1280 ROUNDS=6 TUNES="[None, dict(kernel=5,mt=8,ksplit=1), dict(kernel=5,bm=128), dict(kernel=5,bm=128,pw=16), dict(kernel=5,mt=16,ksplit=2)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_pc.txt
MODE=g128 NBUF=8 MS=512,640,768,1024,1152,1280 ROUNDS=6 TUNES="[None, dict(kernel=5,bm=128), dict(kernel=5,bm=128,pf=4), dict(kernel=5,mt=16,ksplit=2)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_g128.txt
