#!/bin/bash
# round 5, visit 3: the GPU suite on the library with uneven K slices / sms cap / no tiled in the dispatch; then uneven slices in the WIDE kernel's two-slice split
# (1024 tokens per-group is its BASELINE point) and the 256-token choices (128- vs 256-column strips with a skew).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v03; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
T="[None, dict(kernel=5,ksplit=2,skew=-1), dict(kernel=5,ksplit=2,skew=2), dict(kernel=5,ksplit=2,skew=4), dict(kernel=5,ksplit=2,skew=6), dict(kernel=5,ksplit=2,skew=8), dict(kernel=5,ksplit=2,skew=12), dict(kernel=5,bm=128), dict(kernel=5,mt=8)]"
MODE=g128 MS=1024 NBUF=5 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab_wide.txt
MODE=pc MS=1024 NBUF=5 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab_wide.txt
T="[None, dict(kernel=5,bm=128,ksplit=2,skew=-1), dict(kernel=5,bm=128,ksplit=2,skew=3), dict(kernel=5,bm=128,ksplit=2,skew=6), dict(kernel=5,bm=128,ksplit=2,skew=9)]"
MODE=pc MS=384,512 NBUF=5 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab_wide.txt
MODE=g128 MS=512 NBUF=5 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab_wide.txt
T="[None, dict(skew=-1), dict(kernel=4,bm=256,skew=-1), dict(kernel=4,bm=256,skew=2), dict(kernel=4,bm=256,skew=4), dict(kernel=4,bm=128,skew=-1), dict(kernel=4,bm=128,skew=3), dict(kernel=4,bm=256,pw=2,ksplit=2,skew=-1), dict(kernel=4,bm=256,pw=2,ksplit=2,skew=3)]"
MODE=pc MS=192,256 NBUF=5 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab_256.txt
MODE=g128 MS=256 NBUF=5 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab_256.txt
# Llama-2-7B layers at 1024 tokens (configs[3], batch 1): what the wide kernel's shapes and a K split give on 4096 x 4096 / 11008 x 4096 / 4096 x 11008
T="[None, dict(kernel=5), dict(kernel=5,bm=128), dict(kernel=5,mt=8), dict(kernel=5,ksplit=2,skew=-1), dict(kernel=5,ksplit=2,skew=3), dict(kernel=5,bm=128,ksplit=2,skew=-1), dict(kernel=5,bm=128,ksplit=2,skew=3), dict(kernel=5,mt=8,ksplit=2), dict(kernel=4,bm=256,pw=2), dict(kernel=4,bm=256,pw=2,ksplit=2,skew=2)]"
for NK in 4096,4096 11008,4096 4096,11008; do
NK=$NK MODE=pc MS=1024 NBUF=12 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$NK pc   /" | tee -a $O/ab_llama.txt
NK=$NK MODE=g128 MS=1024 NBUF=12 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$NK g128 /" | tee -a $O/ab_llama.txt
done
