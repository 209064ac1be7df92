#!/bin/bash
# Ablation of the 64-column panel shape's loop (profiles/r02_panel_cw2_ablation.txt): builds the operator library with
# -DQQQ_PANEL_ABLATE=<bits> into qqq_amd/libabl_<bits>.so (results of those builds are WRONG by construction) and times them
# against the shipped library on the BASELINE layer.  Run the build part where hipcc is, the timing part on the GPU box:
#   bash tools/ablate_panel.sh build ; gpurun -- 'bash tools/ablate_panel.sh time' ; bash tools/ablate_panel.sh clean
cd "$(dirname "$0")/.."
BITS="1 2 4 8 16 31"
case "$1" in
  build) for b in $BITS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-function -DQQQ_PANEL_ABLATE=$b -o qqq_amd/libabl_$b.so qqq_amd/csrc/qqq_w4a8.hip & done; wait ;;
  time)  L=qqq_amd/libqqq_amd.so; for b in $BITS; do L=$L,qqq_amd/libabl_$b.so; done
         NBUF=1 LIBS=$L ROUNDS=3 ITERS=4 MS=1024,4096 TUNES="[dict(kernel=4,bm=256,mt=8,pw=2)]" python tools/ab.py 2>&1 | grep -v amdgpu.ids
         NBUF=4 LIBS=$L ROUNDS=4 ITERS=4 MS=128 python tools/ab.py 2>&1 | grep -v amdgpu.ids ;;  # the 32-column shape, auto dispatch
  clean) rm -f qqq_amd/libabl_*.so ;;
  *) echo "usage: $0 build|time|clean" ;;
esac
