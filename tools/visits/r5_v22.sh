#!/bin/bash
# round 5, visit 22: the column kernel (now with nt weight loads) against the stream kernel at 8 ... 32 tokens, prefetch depths 2 / 3 / 4, variants repeated; bit-exactness of
# the depths; then smoke + the GPU suite + the driver's bench command on this library
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v22; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 300 python tools/check_variant.py --ms 1,9,16,24,32 --tunes "[dict(kernel=3,pf=2), dict(kernel=3,pf=3), dict(kernel=3,pf=4), dict(kernel=3,mt=2,pf=4), dict(kernel=3,mt=2,pf=8), dict(kernel=4,mt=4)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/check.txt
T="[None, dict(kernel=3,pf=2), dict(kernel=3,pf=3), dict(kernel=3,pf=4), dict(kernel=1), dict(kernel=3,pf=2), dict(kernel=3,pf=3), dict(kernel=3,pf=4), dict(kernel=1)]"
for rep in 1 2; do
NBUF=5 ROUNDS=8 ITERS=4 MS=8,9,12,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab_column_$rep.txt
MODE=g128 NBUF=5 ROUNDS=8 ITERS=4 MS=8,9,12,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab_column_$rep.txt
done
T="[None, dict(kernel=3,mt=2,pf=4), dict(kernel=3,mt=2,pf=8), dict(kernel=1), dict(kernel=3,mt=2,pf=4), dict(kernel=3,mt=2,pf=8), dict(kernel=1)]"
NBUF=5 ROUNDS=8 ITERS=4 MS=17,24,32 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab_column_mt2.txt
MODE=g128 NBUF=5 ROUNDS=8 ITERS=4 MS=17,24,32 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab_column_mt2.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; cp gpurun_out/bench_detail_n1.json $O/bench_a_detail.json
tail -c 400 $O/bench_a.json
