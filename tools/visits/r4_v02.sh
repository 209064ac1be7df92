#!/bin/bash
# round 4, visit 2: first run of the persistent tile walk (tune glds=2): bit-exactness against the tiled kernel, then timing against the one-tile-per-workgroup grid
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v02; mkdir -p $O; export TMPDIR=/tmp
T="[dict(kernel=5,glds=2), dict(kernel=5,glds=2,mt=8), dict(kernel=5,glds=2,bm=128)]"
timeout 600 python tools/check_variant.py --ms 4096,8000 --tunes "$T" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee $O/check.log
timeout 600 python tools/check_variant.py --nk 4096,4096 --ms 8192,9000 --tunes "$T" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee -a $O/check.log
timeout 600 python tools/check_variant.py --nk 11008,4096 --ms 8192 --tunes "$T" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee -a $O/check.log
timeout 600 python tools/check_variant.py --nk 4096,11008 --ms 8192 --tunes "$T" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee -a $O/check.log
timeout 600 python tools/check_variant.py --nk 8384,1152 --ms 4100 --tunes "$T" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee -a $O/check.log
grep -q MISMATCH $O/check.log && echo "MISMATCH found"
TT="[dict(kernel=5,glds=1), dict(kernel=5,glds=2)]"
for nk in 4096,4096 11008,4096 4096,11008; do
  NK=$nk NBUF=8 MS=8192,32768 ROUNDS=5 TUNES="$TT" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_pc.txt
  MODE=g128 NK=$nk NBUF=8 MS=8192 ROUNDS=5 TUNES="$TT" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_g128.txt
done
NBUF=5 MS=4096,8192 ROUNDS=5 TUNES="$TT" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_pc.txt
MODE=g128 NBUF=5 MS=4096 ROUNDS=5 TUNES="$TT" timeout 300 python tools/ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_g128.txt
