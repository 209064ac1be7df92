#!/bin/bash
# round 4, visit 32: the K split of the stream and the panel kernel at 64 / 128 / 256 tokens over eight layer shapes (is "fill 256 workgroups" the right rule on short-K layers?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v32; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
T="[None, dict(kernel=1,ksplit=1), dict(kernel=1,ksplit=2), dict(kernel=1,ksplit=3), dict(kernel=1,ksplit=4), dict(kernel=1,ksplit=6), dict(kernel=1,ksplit=8), dict(kernel=4,ksplit=1), dict(kernel=4,ksplit=2), dict(kernel=4,ksplit=3), dict(kernel=4,ksplit=4), dict(kernel=4,bm=256,ksplit=1), dict(kernel=4,bm=256,ksplit=2)]"
for nk in 8192,3072 3584,3584 4096,4096 5120,5120 11008,4096 8192,8192 18944,3584 14336,4096; do
  for mode in pc g128; do
    NK=$nk MS=64,128,256 MODE=$mode NBUF=12 ROUNDS=5 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$nk $mode /" >> $O/ab.txt
  done
done
wc -l $O/ab.txt
