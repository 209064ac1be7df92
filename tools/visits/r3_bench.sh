#!/bin/bash
# one bench.py run (with its live PMC passes and the cpu_baseline leg) -> gpurun_out/$1/bench.json + a short summary
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-r3bench}; mkdir -p $O; export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > $O/box.txt 2>&1
timeout 1500 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],4), d.get("step_us"))
for k,v in d["per_m"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items()})
for k,v in d["per_m_g128"].items(): print("g128",k, {a:round(b,1) for a,b in v.items()})
print({k:v for k,v in d["roofline"].items() if k not in ("note",)}); print({k:v for k,v in d["roofline_hbm"].items() if k not in ("note",)})
print(d["cpu_baseline"])
PY
