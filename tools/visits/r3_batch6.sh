#!/bin/bash
# round 3, box visit 6: ablation of the wide kernel's loop (M=4096, both modes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b6; mkdir -p $O; export TMPDIR=/tmp
L=qqq_amd/libqqq_amd.so; for b in 1 2 4 8 16 31; do L=$L,qqq_amd/libqqq_amd_abl$b.so; done
LIBS=$L MS=4096 MODE=pc ROUNDS=4 ITERS=4 NBUF=1 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/abl_pc.txt 2>&1; cat $O/abl_pc.txt
LIBS=$L MS=4096 MODE=g128 ROUNDS=3 ITERS=4 NBUF=1 TUNES="[dict(kernel=5)]" timeout 900 python tools/ab.py > $O/abl_g128.txt 2>&1; cat $O/abl_g128.txt
