#!/bin/bash
# round 3, batch 31: deeper LDS-DMA lead (five / six stage buffers, rings of 5 / 10 / 6 / 12 steps): parity, then cold-weight timing
out=gpurun_out/r3b31; mkdir -p $out
timeout 900 python tools/check_variant.py --ms 1000,4096 --tunes "[dict(kernel=5,mt=8,pf=12), dict(kernel=5,mt=8,pf=6), dict(kernel=5,pf=5), dict(kernel=5,pf=10), dict(kernel=5,bm=128,pf=5), dict(kernel=5,bm=128,pf=10,ksplit=2)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tee $out/check.log
grep -q MISMATCH $out/check.log && exit 1
NBUF=8 MS=1024,768 ROUNDS=8 TUNES="[dict(kernel=5,mt=8,pf=4), dict(kernel=5,mt=8,pf=8), dict(kernel=5,mt=8,pf=6), dict(kernel=5,mt=8,pf=12), dict(kernel=5,bm=128,pf=4), dict(kernel=5,bm=128,pf=5), dict(kernel=5,bm=128,pf=10)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_pc.txt
NBUF=8 MS=4096,2048 ROUNDS=6 TUNES="[dict(kernel=5,pf=4), dict(kernel=5,pf=8), dict(kernel=5,pf=5), dict(kernel=5,pf=10)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $out/ab_pc.txt
MODE=g128 NBUF=8 MS=1024,4096 ROUNDS=6 TUNES="[dict(kernel=5,pf=8), dict(kernel=5,pf=5), dict(kernel=5,pf=10), dict(kernel=5,bm=128,pf=8), dict(kernel=5,bm=128,pf=5), dict(kernel=5,bm=128,pf=10)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_g128.txt
