cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bw
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_full_size.py tests/test_wg_kernels.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 > $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c5  $(b c5 4)" >> $O/tests.txt
echo "c5  $(b c5 4)" >> $O/tests.txt
python tools/find_aten_ops.py c5 2>/dev/null | head -8 >> $O/tests.txt
cat $O/tests.txt
