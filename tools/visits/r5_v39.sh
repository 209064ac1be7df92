#!/bin/bash
# round 5, visit 39: the column kernel with sixteen waves per workgroup (tune.waves = 16: the waves split K inside the workgroup) against the eight of the plan, decode regime, cold
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v39; mkdir -p gpurun_out/r5v39; export TMPDIR=/tmp
timeout 200 python tools/check_variant.py --ms 1,7,16 --tunes "[dict(kernel=3,waves=16)]" --ref "dict(kernel=1,ksplit=1)" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/check.txt
T="[None, dict(kernel=3,waves=16), dict(kernel=3), dict(kernel=3,waves=16), None]"
for mode in pc g128; do
MODE=$mode NBUF=0 ROUNDS=8 ITERS=4 MS=1,8,16 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE $mode /" | tee -a $O/ab.txt
for nk in 4096,4096 11008,4096 4096,11008 8192,8192; do
MODE=$mode NK=$nk NBUF=0 ROUNDS=8 ITERS=4 MS=1,8 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$nk $mode /" | tee -a $O/ab.txt
done; done
