# round 4, job k: launch census of the captured c1 / c4 steps (forward + backward + Adam in one hipGraph)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_k
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in c1 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$c -- python $R/bench.py --config $c --hip-graph --step-only --steps 20 --warmup 3 > $O/step_$c.json 2>/dev/null
  f=$(find $O/ks_$c -name "*kernel_stats.csv" | head -1)
  cp $f $O/${c}_graph_kernel_stats.csv
  python $R/tools/kstats.py $f 23 > $O/${c}_graph_kernel_stats_per_step.txt
  find $O/ks_$c -name "*kernel_trace.csv" -delete
done
cd $R
cat $O/step_c1.json; head -45 $O/c1_graph_kernel_stats_per_step.txt
