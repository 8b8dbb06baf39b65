#!/bin/bash
# the weight-gradient stream's HIP priority (GCPNET_SIDE_PRIORITY) against the plain second stream, configs[1] and configs[4] steps, one box
set -u
R=$PWD
OUT=$R/gpurun_out/side_priority.txt
: > $OUT
GCPNET_SIDE_PRIORITY=1 GCPNET_SIDE_PRIORITY_VERBOSE=1 timeout 300 python $R/bench.py --config c2 --step-only --steps 5 --warmup 2 2>&1 | grep "priority" >> $OUT
for rep in 1 2 3; do
for p in "" 1 -1; do
  v=$(GCPNET_SIDE_PRIORITY=$p timeout 300 python $R/bench.py --config c2 --step-only --steps 40 --warmup 10 2>/dev/null | tail -1)
  echo "c2 priority '$p' rep $rep $v" >> $OUT
done
done
for rep in 1 2; do
for p in "" 1 -1; do
  v=$(GCPNET_SIDE_PRIORITY=$p timeout 300 python $R/bench.py --config c5 --step-only --steps 6 --warmup 2 2>/dev/null | tail -1)
  echo "c5 priority '$p' rep $rep $v" >> $OUT
done
done
