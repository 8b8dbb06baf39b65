cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_i
mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 3 --warmup 1 --step-only > $O/c5_step.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
f=$(find $O/ks_c5 -name "*kernel_stats.csv" | head -1); cp $f $O/c5_kernel_stats.csv; python tools/kstats.py $O/c5_kernel_stats.csv 4 | head -30
