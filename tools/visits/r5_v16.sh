#!/bin/bash
# round 5, visit 16: after the ring-depth defaults flipped for the 256 x 256 tiles: automatic vs both forced depths (variants repeated), then smoke, GPU suite, driver bench x2
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v16; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=5,pf=4), dict(kernel=5,pf=8), None, dict(kernel=5,pf=8), dict(kernel=5,pf=4)]"
MS=4096 NBUF=5 ROUNDS=8 ITERS=4 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=4096 NBUF=5 ROUNDS=8 ITERS=4 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
MS=2048,8192 NBUF=3 ROUNDS=6 ITERS=3 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; cp gpurun_out/bench_detail_n1.json $O/bench_a_detail.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b.json 2> $O/bench_b.err
tail -c 600 $O/bench_b.json
bash tools/profile_bench.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
