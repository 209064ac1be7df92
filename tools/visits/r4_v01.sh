#!/bin/bash
# round 4, visit 1 (round-3 build): dispatch check on layer families the cost models were not fitted on (verdict item 8) +
# same-box baseline of the Llama-2-7B linears before the persistent tile walk
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v01; mkdir -p $O; export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; rocm-smi --showclocks 2>/dev/null | head -30; } > $O/box.txt 2>&1
WIDE_SHAPES=1 SHAPES=5120x5120,13824x5120,5120x13824,8192x8192,28672x8192,8192x28672 MS=1,16,64,128,512,2048,8192 ITERS=8 timeout 1200 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids | tee $O/dispatch_check_shapes.txt | tail -5
timeout 900 python tools/bench_llama.py > $O/llama7b.json 2> $O/llama7b.err; echo rc=$?; tail -c 300 $O/llama7b.err
