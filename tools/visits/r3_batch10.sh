#!/bin/bash
# round 3, box visit 10: wide split-K with the batched fold; dispatch check over the mid-m range; whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b10; mkdir -p $O; export TMPDIR=/tmp
MS=768,1024 MODE=pc ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=5, ksplit=2), dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=5, mt=8)]" timeout 900 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
MS=768,1024 MODE=g128 ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=5, ksplit=2), dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=5, mt=8)]" timeout 900 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
SHAPES=8192x21760 MS=384,512,640,768,1024,1152,1280,1536 ITERS=8 timeout 900 python tools/dispatch_check.py > $O/dispatch_base.txt 2>&1; cat $O/dispatch_base.txt
SHAPES=4096x4096,11008x4096,4096x11008 MS=512,1024,2048,4096 ITERS=6 timeout 900 python tools/dispatch_check.py > $O/dispatch_llama.txt 2>&1; cat $O/dispatch_llama.txt
timeout 2400 python -m pytest tests -m gpu -q -x --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -n 12 $O/pytest_gpu.log
