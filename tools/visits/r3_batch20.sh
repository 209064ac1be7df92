#!/bin/bash
# round 3, batch 20: LDS-DMA staging of the wide kernel (all loop loads in asm, hand-counted waits): parity first, then timing
out=gpurun_out/r3b20; mkdir -p $out
timeout 900 python tools/check_variant.py --ms 4096,1024,1000,300,2100 --tunes "[dict(kernel=5), dict(kernel=5,mt=8), dict(kernel=5,pf=8), dict(kernel=5,mt=8,pf=8,ksplit=2), dict(kernel=5,ksplit=3)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | tee $out/check.log
grep -q MISMATCH $out/check.log && exit 1
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=4 MS=4096,2048 ROUNDS=6 TUNES="[dict(kernel=5), dict(kernel=5,pf=8)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_pc.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=8 MS=1024 ROUNDS=6 TUNES="[dict(kernel=5,mt=8), dict(kernel=5,mt=8,pf=8)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_pc_mt8.txt
MODE=g128 LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=4 MS=4096,1024 ROUNDS=6 TUNES="[dict(kernel=5), dict(kernel=5,pf=8)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_g128.txt
