#!/bin/bash
# round 3, box visit 7: uniform slot schedule of the wide kernel vs the first version; ablation of the new loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b7; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_fixtures_all_variants or fused_bias_every_path or pinned" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; tail -4 $O/parity.log
timeout 600 python tools/check_variant.py --ms 4096,1000,257 --tunes "[dict(kernel=5), dict(kernel=5, pf=6), dict(kernel=5, mt=8)]" --ref "dict(kernel=2)" > $O/check_full.log 2>&1; echo "rc=$?" >> $O/check_full.log; cat $O/check_full.log
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_nu.so MS=2048,4096,8192 MODE=pc ROUNDS=5 ITERS=4 TUNES="[dict(kernel=5), dict(kernel=5, pf=6)]" timeout 900 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_nu.so MS=2048,4096 MODE=g128 ROUNDS=4 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_nu.so MS=1024 MODE=pc ROUNDS=4 ITERS=4 TUNES="[dict(kernel=5, mt=8), dict(kernel=4, bm=256, mt=8, pw=2)]" timeout 900 python tools/ab.py > $O/ab_mt8.txt 2>&1; cat $O/ab_mt8.txt
LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_nu.so NK=11008,4096 MS=4096,32768 MODE=pc ROUNDS=3 ITERS=4 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
L=qqq_amd/libqqq_amd.so; for b in 2 16 31; do L=$L,qqq_amd/libqqq_amd_abl$b.so; done
LIBS=$L MS=4096 MODE=pc ROUNDS=4 ITERS=4 NBUF=1 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/abl_pc.txt 2>&1; cat $O/abl_pc.txt
