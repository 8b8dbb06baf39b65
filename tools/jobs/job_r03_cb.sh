cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_cb
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_tn_gemm.py tests/test_side_stream.py tests/test_gpu_parity.py tests/test_wg_kernels.py tests/test_trainer_glue.py -m gpu -q -x 2>&1 | tail -3 > $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c2  $(b c2 20)" >> $O/tests.txt
echo "c5  $(b c5 4)" >> $O/tests.txt
cat $O/tests.txt
