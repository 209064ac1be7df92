#!/bin/bash
# round 5, visit 26: panel kernel, the slices of a tile on ONE XCD whatever the number of strips (tune.fused bit 5 -> hflags 8): layers whose strip count is not a multiple of 8
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v26; mkdir -p gpurun_out/r5v26; export TMPDIR=/tmp
for nk in 11008,4096 13824,5120 4096,11008; do
timeout 300 python tools/check_variant.py --nk $nk --ms 48,64,100,128,300 --tunes "[dict(kernel=4,fused=32), dict(kernel=4,bm=256,fused=32), dict(kernel=4,mt=4,fused=32), dict(kernel=4,ksplit=3,fused=32)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | grep -c "bit-exact" | sed "s/^/$nk bit-exact cases: /" | tee -a $O/check.txt
timeout 300 python tools/check_variant.py --nk $nk --ms 48,64,100,128,300 --tunes "[dict(kernel=4,fused=32), dict(kernel=4,bm=256,fused=32), dict(kernel=4,mt=4,fused=32), dict(kernel=4,ksplit=3,fused=32)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | grep -v "bit-exact" | tail -5 | tee -a $O/check.txt
done
T="[dict(kernel=4), dict(kernel=4,fused=32), dict(kernel=4,bm=256), dict(kernel=4,bm=256,fused=32), dict(kernel=4,mt=4), dict(kernel=4,mt=4,fused=32), dict(kernel=4), dict(kernel=4,fused=32), dict(kernel=4,bm=256), dict(kernel=4,bm=256,fused=32), dict(kernel=1)]"
for nk in 11008,4096 13824,5120 22016,4096 18944,3584 3584,18944 4096,4096 12288,4096; do
NK=$nk NBUF=0 ROUNDS=6 ITERS=4 MS=48,64,128,256 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$nk pc   /" | tee -a $O/ab.txt
done
for nk in 11008,4096 13824,5120; do
MODE=g128 NK=$nk NBUF=0 ROUNDS=6 ITERS=4 MS=64,128 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$nk g128 /" | tee -a $O/ab.txt
done
