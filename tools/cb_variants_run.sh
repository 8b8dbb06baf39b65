#!/bin/bash
# GPU box: times the chain backward of every tools/variants/libgcpnet_hip_c[bf]_*.so at the given tile counts (chain_rows_sweep.py)
R=$PWD; OUT=$R/gpurun_out/${1:-cbx}.txt; shift; T=${1:-4998,2048}
: > $OUT
for L in $(ls $R/tools/variants/libgcpnet_hip_c[bf]_*.so | sort -V); do
  echo "== $(basename $L)" >> $OUT
  GCPNET_HIP_LIB=$L python $R/tools/chain_rows_sweep.py --tiles $T --iters 11 2>&1 | grep tiles | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   tiles', d['tiles'], 'bwd_ms', d['bwd_ms'], 'fwd_ms', d['fwd_ms'])" >> $OUT
done
cat $OUT
