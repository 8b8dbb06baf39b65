#!/bin/bash
# Where the waves of the weight-gradient GEMM kernels spend their cycles: SQ counters of tools/tn_mix_bench.py (one rocprofv3 --pmc pass
# per counter group).  usage (GPU box, repo root): tools/tn_pmc.sh <tag> [mix names ...]
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/${1:-tnpmc}; shift; mkdir -p $OUT; cd /tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/p$i -- python $R/tools/tn_mix_bench.py ${@:-c2 c5} > /dev/null 2>&1
  python $R/tools/pmc_summary.py $OUT/p$i/*/*_counter_collection.csv | grep "tn_"
done
