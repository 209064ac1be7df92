#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b12; mkdir -p $O; export TMPDIR=/tmp
L=qqq_amd/libqqq_amd.so; for b in 1 4 8; do L=$L,qqq_amd/libqqq_amd_abl$b.so; done
LIBS=$L MS=4096 MODE=pc ROUNDS=5 ITERS=4 NBUF=1 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/abl_pc.txt 2>&1; cat $O/abl_pc.txt
LIBS=$L MS=4096 MODE=g128 ROUNDS=4 ITERS=4 NBUF=1 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/abl_g128.txt 2>&1; cat $O/abl_g128.txt
