#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_z; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ks -- python $R/bench.py --config c1 --step-only --steps 4 --warmup 3 > $O/c1.json 2>$O/err.txt
f=$(find $O/ks -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f 7 0 adam_dev_kernel:4 > $O/c1_timeline.txt 2>>$O/err.txt
rm -rf $O/ks
tail -2 $O/c1_timeline.txt
