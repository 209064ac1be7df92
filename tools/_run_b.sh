timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/b_tests.log
tail -3 gpurun_out/b_tests.log
timeout 600 python bench.py --no-cpu > gpurun_out/bench_new.json 2> gpurun_out/bench_new.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_new.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['per_m'].items(): print(k, round(v['us'],1), v['kernel'], v['ksplit'], round(v['roof_frac'],3))
for k,v in d['per_m_g128'].items(): print('g',k, round(v['us'],1))
PY
