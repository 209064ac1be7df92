#!/bin/bash
# rocprofv3 PMC picture of one configuration: tools/pmc_kernel.sh <tag> <M> <mode> '<tune json>' [N,K]
# separate --pmc passes (counters in their own runs, with --kernel-trace only), results under gpurun_out/pmc_<tag>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; TAG=$1; M=$2; MODE=$3; TUNE=$4; NK=${5:-8192,21760}
mkdir -p gpurun_out/pmc_$TAG; cd /tmp; export TMPDIR=/tmp
i=0
for c in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o t -- python $R/tools/prof_calls.py --ms $M --mode $MODE --iters 6 --tune "$TUNE" --nk $NK > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_$TAG > $R/gpurun_out/pmc_$TAG/summary.txt 2>&1
cat $R/gpurun_out/pmc_$TAG/summary.txt
