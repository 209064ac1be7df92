#!/bin/bash
# One GPU-box visit: smoke, the whole -m gpu suite, bench.py (with its live PMC passes).  Everything lands in gpurun_out/r2final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2final; mkdir -p $O
export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > $O/box.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -n 14 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],4), d.get("step_us"))
for k,v in d["per_m"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items()})
for k,v in d["per_m_g128"].items(): print("g128",k, {a:round(b,1) for a,b in v.items()})
print({k:v for k,v in d["roofline"].items() if k not in ("note",)}); print({k:v for k,v in d["roofline_hbm"].items() if k not in ("note",)})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sample"])
PY
