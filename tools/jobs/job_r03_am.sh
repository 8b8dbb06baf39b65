cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_am
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_wg_kernels.py tests/test_gcpnet_equivariance.py -m gpu -q -x 2>&1 | tail -8 > $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c3       $(b c3 10)" >> $O/step.txt
echo "c3       $(b c3 10)" >> $O/step.txt
echo "c2       $(b c2 20)" >> $O/step.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c3 -- python $GRAFT_REPO_ROOT/bench.py --config c3 --step-only --steps 10 --warmup 3 > $O/c3.txt 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
python tools/kstats.py $(find $O/ks_c3 -name "*kernel_stats.csv" | head -1) 13 | head -12 > $O/c3_kstats.txt
cat $O/tests.txt $O/step.txt $O/c3_kstats.txt
