cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ai
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_tn_gemm.py tests/test_gpu_parity.py tests/test_full_size.py tests/test_side_stream.py tests/test_wg_kernels.py -m gpu -q -x 2>&1 | tail -6 > $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c2 x3       $(b c2 20)" >> $O/step.txt
echo "c2 fp32     $(GCPNET_TN_FP32=1 b c2 20)" >> $O/step.txt
echo "c5 x3       $(b c5 4)" >> $O/step.txt
echo "c5 fp32     $(GCPNET_TN_FP32=1 b c5 4)" >> $O/step.txt
cat $O/tests.txt $O/step.txt
