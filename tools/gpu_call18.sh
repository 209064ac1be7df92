#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c18; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 5 $O/pytest_all.log
timeout 900 python tools/bench_llama.py > $O/llama7b.json 2> $O/llama7b.err; tail -c 1500 $O/llama7b.json
