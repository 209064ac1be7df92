#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for st in 0 3 5; do
  if [ $st == 0 ]; then T='{"kernel":2,"bm":258,"glds":2}'; else T="{\"kernel\":2,\"bm\":258,\"glds\":1,\"stages\":$st}"; fi
  bash tools/pmc_kernel.sh g258_s$st 4096 g128 "$T" > /dev/null 2>&1
  echo "#### 258 stages=$st"; grep -E "duration|matrix pipe|LDS bank|SQ_WAIT_INST_LDS /|SQ_LDS_BANK|SQ_LDS_IDX|SQ_INSTS_LDS" gpurun_out/pmc_g258_s$st/summary.txt
done
