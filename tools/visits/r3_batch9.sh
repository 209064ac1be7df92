#!/bin/bash
# round 3, box visit 9: in-launch split-K of the wide kernel -- parity, stress, timing at 768..1280 tokens
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b9; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_fixtures_all_variants or fused_bias_every_path or pinned or wide_inlaunch or baseline_sizes or scratch_stays" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; tail -6 $O/parity.log
timeout 600 python tools/check_variant.py --ms 1024,1000,700 --tunes "[dict(kernel=5, ksplit=2), dict(kernel=5, mt=8, ksplit=2), dict(kernel=5, ksplit=3)]" --ref "dict(kernel=2)" > $O/check_full.log 2>&1; echo "rc=$?" >> $O/check_full.log; grep -c bit-exact $O/check_full.log; grep -v bit-exact $O/check_full.log
MS=640,768,1024,1280 MODE=pc ROUNDS=5 ITERS=4 TUNES="[None, dict(kernel=5, ksplit=2), dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=5, mt=8), dict(kernel=5, ksplit=1)]" timeout 900 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
MS=640,768,1024,1280 MODE=g128 ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=5, ksplit=2), dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=5, ksplit=1)]" timeout 900 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
NK=4096,11008 MS=1024,2048 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5, ksplit=2), dict(kernel=5, ksplit=1)]" timeout 600 python tools/ab.py > $O/ab_llama3.txt 2>&1; cat $O/ab_llama3.txt
NK=11008,4096 MS=1024,2048 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5, ksplit=2), dict(kernel=5, ksplit=1)]" timeout 600 python tools/ab.py > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
SEED=31 SECONDS=40 timeout 300 python tools/fuzz_families.py > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
