cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ao
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in c4 c1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$c -- python $GRAFT_REPO_ROOT/bench.py --config $c --step-only --steps 10 --warmup 3 > $O/$c.txt 2>/dev/null
done
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
for c in c4 c1; do
python tools/kstats.py $(find $O/ks_$c -name "*kernel_stats.csv" | head -1) 13 > $O/${c}_kstats.txt
done
cat $O/c4.txt; head -34 $O/c4_kstats.txt; cat $O/c1.txt; head -5 $O/c1_kstats.txt
