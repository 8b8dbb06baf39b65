cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ax
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/test_wg_kernels.py tests/test_gpu_parity.py tests/test_side_stream.py tests/test_full_size.py -m gpu -q -x 2>&1 | tail -3 > $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c2  $(b c2 20)" >> $O/tests.txt
echo "c4  $(b c4 20)" >> $O/tests.txt
echo "c5  $(b c5 4)" >> $O/tests.txt
cat $O/tests.txt
