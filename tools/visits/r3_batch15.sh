#!/bin/bash
# round 3, batch 15: pair hand-off of the wide kernel's two-slice K split: parity, stress, timing
out=gpurun_out/r3b15; mkdir -p $out
timeout 600 python tools/check_variant.py --ms 1024,768,1000,4096,2100 --tunes "[dict(kernel=5,ksplit=2), dict(kernel=5,mt=8,ksplit=2), dict(kernel=5,ksplit=2,pw=4), dict(kernel=5,ksplit=3)]" 2>&1 | grep -v amdgpu.ids | tee $out/check.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or splitk or stress" 2>&1 | tail -5 | tee $out/pytest.log
NBUF=8 MS=640,768,1024,1152 ROUNDS=6 TUNES="[None, dict(kernel=5,mt=8,ksplit=1), dict(kernel=5,mt=16,ksplit=2), dict(kernel=5,mt=8,ksplit=2)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_pc.txt
MODE=g128 NBUF=8 MS=640,768,1024,1152 ROUNDS=6 TUNES="[None, dict(kernel=5,mt=8,ksplit=1), dict(kernel=5,mt=16,ksplit=2), dict(kernel=5,mt=8,ksplit=2)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_g128.txt
