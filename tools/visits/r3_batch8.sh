#!/bin/bash
# round 3, box visit 8: wide kernel clean-up (uniform schedule only, K % 128 == 0, paired re-quantisations) vs the previous build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b8; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_fixtures_all_variants or fused_bias_every_path or pinned or k_tail" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; tail -4 $O/parity.log
timeout 600 python tools/check_variant.py --ms 4096,1000 --tunes "[dict(kernel=5), dict(kernel=5, pf=6), dict(kernel=5, mt=8)]" --ref "dict(kernel=2)" > $O/check_full.log 2>&1; echo "rc=$?" >> $O/check_full.log; grep -c bit-exact $O/check_full.log; grep -v bit-exact $O/check_full.log
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so MS=2048,4096 MODE=pc ROUNDS=5 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so MS=2048,4096 MODE=g128 ROUNDS=5 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_prev.so NK=11008,4096 MS=8192 MODE=g128 ROUNDS=3 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
