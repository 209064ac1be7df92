#!/bin/bash
# round 4, visit 6: smoke, the whole -m gpu suite, bench.py (with the new llama7b / clocks / eager objects)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v06; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -n 16 $O/pytest_gpu.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],4), d.get("step_us"), "eager", d.get("eager"))
for k,v in d["per_m"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items()})
for k,v in d["per_m_g128"].items(): print("g128",k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items()})
print({k:v for k,v in d["roofline"].items() if k not in ("note",)}); print(d.get("clocks"))
l=d.get("llama7b",{}); print("llama seconds", l.get("seconds"), "skipped", l.get("skipped")); print(json.dumps(l.get("sum_of_7_linears")), json.dumps(l.get("sum_of_4_merged_linears")))
for mode,v in l.get("layers",{}).items():
    for name,x in v.items(): print(mode,name,{m:(round(y["gemm_tops"]),round(y["speedup_vs_fp16"],2)) for m,y in x.items()})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["torch_fp16_cpu_gemm"])
PY
