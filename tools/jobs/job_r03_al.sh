cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_al
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c3 -- python $GRAFT_REPO_ROOT/bench.py --config c3 --step-only --steps 10 --warmup 3 > $O/c3.txt 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
python tools/kstats.py $(find $O/ks_c3 -name "*kernel_stats.csv" | head -1) 13 | head -30 > $O/c3_kstats.txt
cat $O/c3.txt $O/c3_kstats.txt
