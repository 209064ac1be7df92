#!/bin/bash
# round 4, visit 11: whole -m gpu suite on the current build; dispatch check around the split points (interleaved); bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v11; mkdir -p $O; rm -f $O/*; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -n 10 $O/pytest_gpu.log
WIDE_SHAPES=1 SHAPES=8192x21760 MS=320,512,640,768,1024,1280,1536,2048,3072 ITERS=9 timeout 900 python tools/dispatch_check.py 2>&1 | grep -v amdgpu.ids | tee $O/dispatch_check_mid.txt | cut -c1-60,200-420
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],4), "eager", d.get("eager",{}).get("value"))
for k,v in d["per_m"].items(): print(k, v["kernel"], v["ksplit"], round(v["us_median"],1), round(v["roof_frac_median"],3))
for k,v in d["per_m_g128"].items(): print("g128",k, v["kernel"], v["ksplit"], round(v["us_median"],1), round(v["roof_frac_median"],3))
print(d["roofline"]["frac"], d.get("clocks",{}).get("busy_sclk",{}).get("mhz"))
l=d.get("llama7b",{}); print(json.dumps(l.get("sum_of_7_linears")))
PY
