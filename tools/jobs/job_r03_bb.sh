cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --config c5 --step-only --steps 2 --warmup 1 > $O/c5.txt 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 3 100 > $O/c5_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
head -120 $O/c5_timeline.txt
