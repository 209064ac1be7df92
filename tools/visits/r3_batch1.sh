#!/bin/bash
# round 3, box visit 1: the wide kernel (correctness on goldens + full size, then timing), static-priority panel build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b1; mkdir -p $O; export TMPDIR=/tmp
W='[dict(kernel=5), dict(kernel=5, pf=3), dict(kernel=5, stages=3), dict(kernel=5, pw=4)]'
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_fixtures_all_variants or fused_bias_every_path" > $O/golden.log 2>&1; echo "rc=$?" >> $O/golden.log; tail -5 $O/golden.log
timeout 600 python tools/check_variant.py --ms 4096,1000,257 --tunes "$W" > $O/check_full.log 2>&1; echo "rc=$?" >> $O/check_full.log; cat $O/check_full.log
timeout 300 python tools/check_variant.py --nk 11008,4096 --ms 2048,300 --tunes "$W" > $O/check_llama.log 2>&1; echo "rc=$?" >> $O/check_llama.log; tail -12 $O/check_llama.log
timeout 300 python tools/check_variant.py --nk 4160,4160 --ms 700 --tunes "$W" --ref "dict(kernel=1)" > $O/check_ragged.log 2>&1; echo "rc=$?" >> $O/check_ragged.log; tail -6 $O/check_ragged.log
MS=1024,2048,4096,8192 MODE=pc ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=5), dict(kernel=5, pf=3), dict(kernel=5, stages=3), dict(kernel=5, pw=4), dict(kernel=5, pw=16), dict(kernel=2)]" timeout 600 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
MS=1024,2048,4096 MODE=g128 ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=5), dict(kernel=5, pf=3), dict(kernel=5, stages=3), dict(kernel=4, bm=256, mt=8, pw=2)]" timeout 600 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prio.so MS=128,1024,4096 MODE=pc ROUNDS=5 ITERS=4 timeout 300 python tools/ab.py > $O/ab_prio.txt 2>&1; cat $O/ab_prio.txt
NK=11008,4096 MS=2048,4096,8192 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5), dict(kernel=2)]" timeout 300 python tools/ab.py > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
