#!/bin/bash
# round 5, visit 5: the panel kernel's FEEDER variant (tune.waves = 9: a ninth wave stages the activations by LDS-DMA, the eight compute waves load weights only):
# bit-exactness (ragged m, K % 128 == 64, 1 ... 4 slices with and without skew, both modes), then timing against the plain kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v05; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
V="[dict(kernel=4,waves=9), dict(kernel=4,waves=9,ksplit=1), dict(kernel=4,waves=9,ksplit=2,skew=-1), dict(kernel=4,waves=9,ksplit=3,skew=7), dict(kernel=4,waves=9,mt=4), dict(kernel=4,waves=9,mt=4,ksplit=2)]"
timeout 120 python tools/check_variant.py --ms 33,64,65,100,128,200,300 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
timeout 120 python tools/check_variant.py --nk 4096,4160 --modes pc --ms 40,128,129 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
timeout 120 python tools/check_variant.py --nk 4096,4096 --ms 128,1000 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
timeout 120 python tools/check_variant.py --nk 1088,512 --ms 70,128 --tunes "$V" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
T="[None, dict(kernel=4), dict(kernel=4,waves=9), dict(kernel=4,waves=9,skew=-1), dict(kernel=4,waves=9,skew=6), dict(kernel=4,waves=9,ksplit=2), dict(kernel=4,mt=4,ksplit=4), dict(kernel=4,mt=4,ksplit=4,waves=9)]"
MS=64,96,128,256 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
MODE=g128 MS=64,128,256 NBUF=5 ROUNDS=8 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
T="[None, dict(kernel=4), dict(kernel=4,waves=9), dict(kernel=4,mt=4), dict(kernel=4,mt=4,waves=9)]"
for NK in 4096,4096 11008,4096 4096,11008 8192,8192; do
NK=$NK MS=64,128,256,512,1024 NBUF=12 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$NK pc   /" | tee -a $O/ab_shapes.txt
done
NK=4096,4096 MODE=g128 MS=128,1024 NBUF=12 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/4096,4096 g128 /" | tee -a $O/ab_shapes.txt
