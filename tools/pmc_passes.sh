#!/bin/bash
# rocprofv3 PMC passes over one bench step (run on the GPU box through gpurun).
# usage: tools/pmc_passes.sh <out-prefix under gpurun_out/> [bench args...]
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-in-library-multi --sustain-s 0 --no-pmc "$@" > $R/gpurun_out/$OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/$OUT/p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/$OUT -name '*counter_collection.csv') > $R/gpurun_out/$OUT/summary.txt
# keep only the summaries (the raw csv files are large)
find $R/gpurun_out/$OUT -name '*.csv' -size +2M -delete
head -60 $R/gpurun_out/$OUT/summary.txt
