#!/bin/bash
# HBM traffic of the c2 step's kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (GPU box, repo root)
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/${1:-qp}; mkdir -p $OUT; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python $R/bench.py --steps 3 --warmup 1 --step-only > /dev/null 2>&1
  python $R/tools/pmc_summary.py $OUT/pmc_$C/*/*_counter_collection.csv | grep -i "chain_bwd\|wg_fwd_kernel<4, 1, true, 1>"
done
