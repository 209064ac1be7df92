#!/bin/bash
# round 3, batch 27: direct (LDS-free) epilogue of the wide kernel: parity (with and without bias: pytest), then timing
out=gpurun_out/r3b27; mkdir -p $out
timeout 900 python tools/check_variant.py --ms 4096,1024,1000,512 --tunes "[dict(kernel=5), dict(kernel=5,mt=8), dict(kernel=5,bm=128), dict(kernel=5,pf=8,pw=4)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tee $out/check.log
grep -q MISMATCH $out/check.log && exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_qlinear.py -q -x 2>&1 | tail -4 | tee $out/pytest.log
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=4 MS=4096,2048 ROUNDS=6 TUNES="[dict(kernel=5)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_pc.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=8 MS=1024,768 ROUNDS=6 TUNES="[dict(kernel=5,mt=8), dict(kernel=5,bm=128)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $out/ab_pc.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=8 NK=4096,4096 MS=4096,2048 ROUNDS=6 TUNES="[None]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_llama.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=4 NK=11008,4096 MS=4096,8192 ROUNDS=6 TUNES="[None]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $out/ab_llama.txt
MODE=g128 LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NBUF=4 MS=4096,1024 ROUNDS=6 TUNES="[None]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_g128.txt
