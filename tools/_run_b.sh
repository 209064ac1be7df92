cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/prof_bench; rm -rf $O; mkdir -p $O
( cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu --no-fp16 --no-pmc --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python $R/tools/kernel_trace_table.py $O/trace > $O/bench_kernels_by_shape.csv
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null )
cut -c1-200 $O/bench_kernels_by_shape.csv | head -30
cd $R
bash tools/pmc_kernel.sh panel_m128 128 pc '{}' > /dev/null 2>&1
bash tools/pmc_kernel.sh panel_m4096 4096 pc '{}' > /dev/null 2>&1
for t in panel_m128 panel_m4096; do echo "#### $t"; grep -v "^   [A-Z]" gpurun_out/pmc_$t/summary.txt; done
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d['per_m'].items(): print(k, round(v['us'],1), v['kernel'], v['ksplit'], round(v['roof_frac'],3))
for k,v in d['per_m_g128'].items(): print('g',k, round(v['us'],1))
PY
