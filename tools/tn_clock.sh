#!/bin/bash
# Effective shader clock of the weight-gradient GEMM kernels: GRBM_GUI_ACTIVE / kernel duration (rocprofv3, separate runs for
# the counter and the trace).  usage (GPU box, repo root): tools/tn_clock.sh <tag> [env assignments ...]
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/${1:-clk}; shift; mkdir -p $OUT; cd /tmp
env "$@" rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python $R/tools/tn_mix_bench.py c2 c5 > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT/pmc/*/*_counter_collection.csv
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/tools/tn_mix_bench.py c2 c5 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v.sort()
    print(f"{k:60s} n={len(v):4d} median {v[len(v)//2]/1e3:9.1f} us")
PY
