timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/b_tests.log
tail -3 gpurun_out/b_tests.log
QQQ_AMD_LIB=qqq_amd/libtrace.so MS=128 timeout 300 python tools/trace_panel.py 2>&1 | grep -v amdgpu.ids > gpurun_out/trace_pc4.log
grep "finisher:\|launches" gpurun_out/trace_pc4.log
NBUF=4 MODE=pc LIBS=qqq_amd/libbase2.so,qqq_amd/libqqq_amd.so ROUNDS=6 ITERS=4 MS=64,128,256,512 timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b_ab_pc.log
cat gpurun_out/b_ab_pc.log
NBUF=4 MODE=pc LIBS=qqq_amd/libqqq_amd.so ROUNDS=5 ITERS=6 MS=16,32 TUNES="[None,dict(kernel=4,mt=1),dict(kernel=4,mt=1,pf=8),dict(kernel=4,mt=1,bm=256),dict(kernel=4,mt=2),dict(kernel=4,mt=2,pf=8),dict(kernel=4,mt=2,bm=256),dict(kernel=4,mt=1,ksplit=3),dict(kernel=4,mt=2,ksplit=3)]" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/b_ab_m16.log
cat gpurun_out/b_ab_m16.log
