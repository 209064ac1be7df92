#!/bin/bash
# round 5, visit 27: EVERY dispatch grid again on the final library (nt weight loads in the column kernel / 64-token m-blocks), points of up to 128 tokens measured COLD (1.1 GB of rotating weight copies per layer, the rotation continuing from group to group)
# (library: + the slices of a tile on one XCD in the panel kernel, + the panel kernel as a candidate at 9 ... 32 tokens); then smoke, the GPU suite, the driver's bench command with 12 rotating weight copies
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v27; mkdir -p gpurun_out/r5v27; export TMPDIR=/tmp
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
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; cp gpurun_out/bench_detail_n1.json $O/bench_a_detail.json
tail -c 300 $O/bench_a.json
