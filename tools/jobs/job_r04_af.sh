#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_af; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --config c3 --step-only --steps 10 --warmup 3 > $O/c3.json 2>$O/err.txt
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
python $R/tools/kstats.py $f 10 > $O/c3_kernel_stats_per_step.txt 2>>$O/err.txt
f=$(find $O/ks -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f 13 0 adam_dev_kernel:8 > $O/c3_timeline.txt 2>>$O/err.txt
rm -rf $O/ks
cd $R
timeout 600 python tools/host_enqueue_time.py c3 20 > $O/host_c3.txt 2>>$O/err.txt
cat $O/host_c3.txt; tail -1 $O/c3_timeline.txt; head -40 $O/c3_kernel_stats_per_step.txt
