cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bm
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_wg_kernels.py tests/test_full_size.py tests/test_side_stream.py tests/test_gcpnet_equivariance.py tests/test_trainer_glue.py -m gpu -q -x 2>&1 | tail -6 > $O/tests.txt
SWEEP_POISON=1 SWEEP_REPEAT=2 timeout 900 python tests/sweep_layers.py 60 51 2>&1 | tail -1 >> $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for i in 1 2; do
echo "c2 fused agg     $(b c2 20)" >> $O/tests.txt
echo "c2 separate agg  $(GCPNET_FUSE_AGG=0 b c2 20)" >> $O/tests.txt
done
echo "c3 fused agg     $(b c3 10)" >> $O/tests.txt
echo "c3 separate agg  $(GCPNET_FUSE_AGG=0 b c3 10)" >> $O/tests.txt
echo "c5 fused agg     $(b c5 4)" >> $O/tests.txt
echo "c5 separate agg  $(GCPNET_FUSE_AGG=0 b c5 4)" >> $O/tests.txt
cat $O/tests.txt
