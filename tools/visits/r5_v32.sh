#!/bin/bash
# round 5, visit 32: the M split the refitted rates now plan at 4700 ... 5120 tokens on the BASELINE layer (4096 + the rest) against the one launch; then the final validation
# of the round's last library: smoke, the GPU suite, the driver's bench command twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v32; mkdir -p gpurun_out/r5v32; export TMPDIR=/tmp
T="[None, dict(split_m=-1), None, dict(split_m=-1)]"
NBUF=5 ROUNDS=6 ITERS=4 MS=4700,5000,5120,2900 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 NBUF=5 ROUNDS=6 ITERS=4 MS=5000 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
T="[None, dict(kernel=5,ksplit=2), dict(kernel=5,bm=128), None, dict(kernel=5,ksplit=2), dict(kernel=5,bm=128)]"
NBUF=5 ROUNDS=8 ITERS=4 MS=768,1024,1280 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab_m1024.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; cp gpurun_out/bench_detail_n1.json $O/bench_a_detail.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b.json 2> $O/bench_b.err
tail -c 300 $O/bench_b.json
