#!/bin/bash
# configs[4] step against the weight-gradient GEMM's target split count (GCPNET_TN_SPLITS; default 128), one box: gpurun_out/c5_splits.txt
set -u
R=$PWD
OUT=$R/gpurun_out/c5_splits.txt
: > $OUT
for rep in 1 2; do
for s in 128 108 96 146 72 64 192; do
  v=$(GCPNET_TN_SPLITS=$s timeout 300 python $R/bench.py --config c5 --step-only --steps 6 --warmup 2 2>/dev/null | tail -1)
  echo "splits $s rep $rep $v" >> $OUT
done
done
