#!/bin/bash
# Profile passes of one round (run on the GPU box from the repo root): bench line, rocprofv3 kernel stats of the same command,
# PMC passes (separate runs, no tracing).  usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- python $R/bench.py --no-cpu-baseline --no-c5-block --no-other-configs > $OUT/bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c5 -- python $R/bench.py --config c5 --steps 3 --warmup 1 --step-only > $OUT/c5_step.json 2> /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python $R/bench.py --steps 3 --warmup 1 --step-only > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python $R/bench.py --steps 3 --warmup 1 --step-only > /dev/null 2>&1
# the same three passes on the configs[4] step (its dominant kernel: the (256,32) block backward)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_c5 -- python $R/bench.py --config c5 --steps 2 --warmup 1 --step-only > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_c5 -- python $R/bench.py --config c5 --steps 2 --warmup 1 --step-only > /dev/null 2>&1
cd $R
find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
# (the traces themselves are large: keep only the stats / counter tables)
find $OUT -name "*kernel_trace.csv" -delete
# condensed tables (what gets copied into profiles/)
{
  for C in FETCH_SIZE WRITE_SIZE mfma FETCH_SIZE_c5 WRITE_SIZE_c5 mfma_c5; do
    f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
    echo "== rocprofv3 --pmc $C (mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB; _c5: the configs[4] step)"
    python tools/pmc_summary.py $f
  done
} > $OUT/pmc_summary.txt 2>&1
cp $(find $OUT/ks -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/ks_c5 -name "*kernel_stats.csv" | head -1) $OUT/c5_kernel_stats.csv
python tools/kstats.py $OUT/kernel_stats.csv 25 > $OUT/kernel_stats_per_step.txt
python tools/kstats.py $OUT/c5_kernel_stats.csv 4 > $OUT/c5_kernel_stats_per_step.txt
python tools/make_traffic.py $OUT $OUT/traffic.json > /dev/null 2>&1
# (resource usage of the big kernels -- VGPRs, spills, occupancy: `python tools/kres.py gcpnet_amd/csrc/<file>.hip`, run where hipcc is
# cheap, not on the GPU box; committed as profiles/<tag>_kernel_resource_usage.txt)
find $OUT -name "*counter_collection.csv" -size +8M -delete
