QQQ_AMD_LIB=qqq_amd/librelax.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or baseline or stress or splitk or load" 2>&1 | tail -3
for mode in pc g128; do
NBUF=4 MODE=$mode LIBS=qqq_amd/libqqq_amd.so,qqq_amd/librelax.so ROUNDS=6 ITERS=4 MS=64,96,128,192 python tools/ab.py 2>&1 | grep -v amdgpu.ids
done
NK=4096,4096 NBUF=8 MODE=pc LIBS=qqq_amd/libqqq_amd.so,qqq_amd/librelax.so ROUNDS=6 ITERS=4 MS=256,512,1024 python tools/ab.py 2>&1 | grep -v amdgpu.ids
NK=4096,11008 NBUF=8 MODE=pc LIBS=qqq_amd/libqqq_amd.so,qqq_amd/librelax.so ROUNDS=6 ITERS=4 MS=256,512,1024 python tools/ab.py 2>&1 | grep -v amdgpu.ids
