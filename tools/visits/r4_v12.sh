#!/bin/bash
# round 4, visit 12: the tile walk's flush through a wave-private LDS patch (-DQQQ_WIDE_FLUSH_LDS=1 build) against the row-swap flush
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v12; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_flds.so timeout 600 python tools/check_variant.py --nk 4096,4096 --ms 8192,9000 --tunes "[dict(kernel=5,glds=2)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/check.log
QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_flds.so timeout 600 python tools/check_variant.py --nk 8384,1152 --ms 4100 --tunes "[dict(kernel=5,glds=2)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
T="[dict(kernel=5,glds=1), dict(kernel=5,glds=2)]"
for nk in 4096,4096 11008,4096; do
  LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_flds.so NK=$nk NBUF=8 MS=8192,16384,32768 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$nk pc /" | tee -a $O/ab.txt
  LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_flds.so MODE=g128 NK=$nk NBUF=8 MS=8192,32768 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$nk g128 /" | tee -a $O/ab.txt
done
