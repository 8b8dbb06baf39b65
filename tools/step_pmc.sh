#!/bin/bash
# Where the waves of a step's big kernels spend their cycles: SQ counters of `bench.py --step-only` (one rocprofv3 --pmc pass per group).
# usage (GPU box, repo root): tools/step_pmc.sh <tag> [bench.py arguments ...]      e.g. tools/step_pmc.sh sq_c2 --steps 3 --warmup 1
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/${1:-sq}; shift; mkdir -p $OUT; cd /tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
         "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/p$i -- python $R/bench.py --step-only "$@" > /dev/null 2>&1
  python $R/tools/pmc_summary.py $OUT/p$i/*/*_counter_collection.csv
done
