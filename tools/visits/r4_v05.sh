#!/bin/bash
# round 4, visit 5: the tile walk's parity test, then plain vs tile walk over tokens x layer shapes x modes (for the dispatch rule), then the seam timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v05; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile_walk or wide_every" 2>&1 | tail -5 | tee $O/pytest.txt
TT="[dict(kernel=5,glds=1), dict(kernel=5,glds=2), dict(kernel=5,glds=1,bm=128), dict(kernel=5,glds=2,bm=128), dict(kernel=5,glds=1,mt=8), dict(kernel=5,glds=2,mt=8)]"
for nk in 4096,4096 11008,4096 4096,11008 12288,4096 22016,4096 8192,8192 5120,5120 13824,5120 8192,21760; do
  NK=$nk NBUF=6 MS=2048,4096,8192,16384,32768 ROUNDS=4 TUNES="$TT" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$nk pc /" | tee -a $O/ab.txt
  MODE=g128 NK=$nk NBUF=6 MS=2048,4096,8192,16384,32768 ROUNDS=4 TUNES="$TT" timeout 400 python tools/ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NK=$nk g128 /" | tee -a $O/ab.txt
done
export QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_trace.so
NK=4096,4096 MS=8192,32768 timeout 300 python tools/trace_chain.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_chain.txt
MODE=g128 NK=4096,4096 MS=8192 timeout 300 python tools/trace_chain.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_chain.txt
