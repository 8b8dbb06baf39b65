#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ak; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ks -- python $R/bench.py --config c2 --step-only --steps 6 --warmup 3 > $O/prof_bench.txt 2>$O/err.txt
f=$(find $O/ks -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f 9 0 gcp2_chain_fwd_kernel:4 > $O/c2_timeline.txt 2>>$O/err.txt
rm -rf $O/ks
tail -1 $O/c2_timeline.txt
