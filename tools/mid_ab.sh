#!/bin/bash
# Same-box A/B of the five-wave GEMM form + fused gate gradients (GCPNET_TN_MID=1 / 0): kernel stats of the configs[1] bench under
# rocprofv3 (run on the GPU box from the repo root) -> gpurun_out/mid_ab/
set -u
R=$PWD
OUT=$R/gpurun_out/mid_ab
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for m in 1 0; do
  GCPNET_TN_MID=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks$m -- python $R/bench.py --no-cpu-baseline --no-c5-block --no-other-configs > $OUT/bench_$m.json 2> /dev/null
  f=$(find $OUT/ks$m -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then (cd $R && python tools/kstats.py $f 25 > $OUT/kstats_$m.txt); fi
  find $OUT/ks$m -name "*.csv" -delete
done
for i in 1 2 3; do for m in 1 0; do
  GCPNET_TN_MID=$m timeout 300 python $R/bench.py --no-cpu-baseline --no-c5-block --config c3 --step-only --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/c3_${m}_$i.json
done; done
