#!/bin/bash
# round 3, box visit 5: 128-token wide tiles (mt = 8)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b5; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_fixtures_all_variants or fused_bias_every_path" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; tail -4 $O/parity.log
timeout 600 python tools/check_variant.py --ms 1024,1000,129 --tunes "[dict(kernel=5, mt=8), dict(kernel=5, mt=8, pf=6), dict(kernel=5, mt=8, stages=3)]" > $O/check_full.log 2>&1; echo "rc=$?" >> $O/check_full.log; cat $O/check_full.log
MS=512,768,1024,1280,2048,4096 MODE=pc ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=5, mt=8), dict(kernel=5, mt=8, pf=6), dict(kernel=5, mt=8, stages=3), dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
MS=1024,2048 MODE=g128 ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5, mt=8), dict(kernel=4, bm=256, mt=8, pw=2)]" timeout 600 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
NK=4096,4096 MS=1024,2048,4096 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5, mt=8), dict(kernel=5)]" timeout 300 python tools/ab.py > $O/ab_llama2.txt 2>&1; cat $O/ab_llama2.txt
NK=11008,4096 MS=1024,2048,4096 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5, mt=8), dict(kernel=5)]" timeout 300 python tools/ab.py > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
NK=4096,11008 MS=1024,2048,4096 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5, mt=8), dict(kernel=5)]" timeout 300 python tools/ab.py > $O/ab_llama3.txt 2>&1; cat $O/ab_llama3.txt
