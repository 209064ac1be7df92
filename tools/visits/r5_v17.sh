#!/bin/bash
# round 5, visit 17: what one CU can pull through its vector memory path when all 256 do the same (tools/l2_fill_bench.hip): own L2 window, shared activation panel
# in lockstep / rotated, own HBM stream, and the 128-token loop's mix -- the ceiling the panel kernel's 43-53 KB/us per CU is to be read against
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5v17; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/l2_fill_bench.hip -o /tmp/l2fb 2>/dev/null || exit 1
timeout 120 /tmp/l2fb 16 | tee $O/l2_fill.txt
timeout 120 /tmp/l2fb 16 | tee $O/l2_fill_again.txt
# the per-group two-slice split of the 256 x 256 tiles (384 ... 1024 tokens): does the new ring depth (4) hold there too?
T="[None, dict(kernel=5,ksplit=2,pf=4), dict(kernel=5,ksplit=2,pf=8), None, dict(kernel=5,ksplit=2,pf=8), dict(kernel=5,ksplit=2,pf=4)]"
MODE=g128 MS=512,1024 NBUF=5 ROUNDS=8 ITERS=4 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE g128 /" | tee -a $O/ab.txt
MS=768 NBUF=5 ROUNDS=8 ITERS=4 TUNES="$T" timeout 600 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/BASELINE pc   /" | tee -a $O/ab.txt
