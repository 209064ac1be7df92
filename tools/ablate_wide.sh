#!/bin/bash
# Where the wide kernel's loop time goes: operator-library builds with parts of the steady-state loop removed
# (-DQQQ_WIDE_ABLATE=<bits>, results wrong by construction) -> qqq_amd/libqqq_amd_abl<bits>.so; then, on the GPU box:
#   LIBS=qqq_amd/libqqq_amd.so,qqq_amd/libqqq_amd_abl1.so,... TUNES="[dict(kernel=5)]" MS=4096 python tools/ab.py
cd "$(dirname "$0")/.."
for b in ${BITS:-1 2 4 8 16 31}; do
  ( QQQ_AMD_LIB=$PWD/qqq_amd/libqqq_amd_abl$b.so QQQ_AMD_CXXFLAGS=-DQQQ_WIDE_ABLATE=$b python -c "from qqq_amd import build as kb; print(kb.build(force=True))" ) &
  if (( $(jobs -r | wc -l) >= 3 )); then wait -n; fi
done
wait
