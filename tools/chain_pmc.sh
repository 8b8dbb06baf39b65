#!/bin/bash
# SQ / instruction-cache / clock counters of the chain kernels alone (one rocprofv3 --pmc pass per group; no tracing flags).
# usage (GPU box, repo root): tools/chain_pmc.sh <tag> [tiles]
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/${1:-cpmc}; mkdir -p $OUT; cd /tmp
T=${2:-4998}
i=0
for G in "GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/p$i -- python $R/tools/chain_rows_sweep.py --tiles $T --iters 3 > $OUT/p$i.log 2>&1
  python $R/tools/pmc_summary.py $OUT/p$i/*/*_counter_collection.csv | grep -i "chain" | tee -a $OUT/summary.txt
done
