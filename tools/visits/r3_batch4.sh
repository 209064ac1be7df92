#!/bin/bash
# round 3, box visit 4: wide epilogue (branch-free interior path) -- timeline, timing, parity on the goldens (guarded path)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b4; mkdir -p $O; export TMPDIR=/tmp
T=$PWD/qqq_amd/libqqq_amd_trace.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_fixtures_all_variants or fused_bias_every_path or pinned or baseline_sizes" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; tail -4 $O/parity.log
QQQ_AMD_LIB=$T MS=4096 timeout 300 python tools/trace_wide.py > $O/trace_pc.txt 2>&1; grep -E "==|prologue|main loop|epilogue|kernel end" $O/trace_pc.txt
MS=2048,4096,8192 MODE=pc ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=4, bm=256, mt=8, pw=2)]" timeout 600 python tools/ab.py > $O/ab_pc.txt 2>&1; cat $O/ab_pc.txt
MS=2048,4096 MODE=g128 ROUNDS=4 ITERS=4 TUNES="[None, dict(kernel=2)]" timeout 600 python tools/ab.py > $O/ab_g128.txt 2>&1; cat $O/ab_g128.txt
NK=11008,4096 MS=4096,32768 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5), dict(kernel=2)]" timeout 300 python tools/ab.py > $O/ab_llama.txt 2>&1; cat $O/ab_llama.txt
NK=4096,4096 MS=4096,8192 MODE=pc ROUNDS=3 ITERS=4 TUNES="[None, dict(kernel=5), dict(kernel=2)]" timeout 300 python tools/ab.py > $O/ab_llama2.txt 2>&1; cat $O/ab_llama2.txt
