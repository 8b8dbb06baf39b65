# round 4, job s: segment_reduce with eight rows in flight: tests (bitwise comparisons between routes included), c2 / c1 / c5 steps
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_side_stream.py tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | tail -2 > $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 $3 2>/dev/null; }
echo "c2 $(b c2 20)" >> $O/step.txt
echo "c2 $(b c2 20)" >> $O/step.txt
echo "c1 graph $(b c1 30 --hip-graph)" >> $O/step.txt
echo "c5 $(b c5 4)" >> $O/step.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $GRAFT_REPO_ROOT/bench.py --step-only --steps 20 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $(find $O/ks -name "*kernel_stats.csv" | head -1) 25 | grep -i "segment_reduce\|total" >> $O/step.txt
find $O/ks -name "*kernel_trace.csv" -delete
cat $O/tests.txt $O/step.txt
