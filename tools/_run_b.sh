SEED=11 SECONDS=90 timeout 200 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/fuzz11.log; cat gpurun_out/fuzz11.log
SEED=12 SECONDS=60 timeout 200 python tools/fuzz_families.py 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/fuzz12.log; cat gpurun_out/fuzz12.log
