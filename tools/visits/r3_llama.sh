#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3llama; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python tools/bench_llama.py > $O/llama7b.json 2> $O/llama7b.err; echo rc=$?; tail -c 600 $O/llama7b.err; python -c "
import json; d=json.load(open('$O/llama7b.json')); print(json.dumps(d)[:3000])"
