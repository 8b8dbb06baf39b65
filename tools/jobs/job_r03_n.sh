cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_n
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_wg_kernels.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4 > $O/tests.txt
c5() { python bench.py --config c5 --step-only --steps 3 --warmup 2 2>/dev/null; }
echo "c5 head4      $(c5)" > $O/c5.txt
echo "c5 head8      $(GCPNET_WG_FWD_HEAD4=0 c5)" >> $O/c5.txt
echo "c5 head4      $(c5)" >> $O/c5.txt
echo "c5 head8      $(GCPNET_WG_FWD_HEAD4=0 c5)" >> $O/c5.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 3 --warmup 1 --step-only > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
python tools/kstats.py $(find $O/ks_c5 -name "*kernel_stats.csv" | head -1) 4 | head -16 > $O/c5_kstats.txt
cat $O/tests.txt $O/c5.txt $O/c5_kstats.txt
