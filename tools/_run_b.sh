timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/b_tests.log
tail -3 gpurun_out/b_tests.log
for mode in pc g128; do
NBUF=4 MODE=$mode LIBS=qqq_amd/libbase.so,qqq_amd/libqqq_amd.so ROUNDS=6 ITERS=4 MS=128,256,512,1024,4096 timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b_ab_$mode.log
cat gpurun_out/b_ab_$mode.log
done
