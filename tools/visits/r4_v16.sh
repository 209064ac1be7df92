#!/bin/bash
# round 4, visit 16: cache policy of the tile walk's D stores (-DQQQ_WIDE_FLUSH_AUX builds: 0 write-back (shipped), 2 nt, 17 sc0 sc1 write-through, 19 all)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v16; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
for a in 2 17 19; do
  QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_fa$a.so timeout 300 python tools/check_variant.py --nk 4096,4096 --ms 8192,9000 --tunes "[dict(kernel=5,glds=2), dict(kernel=5,glds=2,bm=128)]" --ref "dict(kernel=2)" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/check.log
done
T="[dict(kernel=5,glds=2)]"
L=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_fa2.so,qqq_amd/libqqq_amd_fa17.so,qqq_amd/libqqq_amd_fa19.so
for nk in 4096,4096 11008,4096; do
  LIBS=$L NK=$nk NBUF=8 MS=8192,32768 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$nk pc /" | tee -a $O/ab.txt
  LIBS=$L MODE=g128 NK=$nk NBUF=8 MS=8192 ROUNDS=6 TUNES="$T" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$nk g128 /" | tee -a $O/ab.txt
done
