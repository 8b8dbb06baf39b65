cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bh
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_wg_kernels.py tests/test_gpu_parity.py tests/test_full_size.py -m gpu -q -x 2>&1 | tail -3 > $O/tests.txt
SWEEP_POISON=1 timeout 900 python tests/sweep_gcp2.py 150 41 2>&1 | tail -1 >> $O/tests.txt
SWEEP_POISON=1 timeout 900 python tests/sweep_layers.py 40 42 2>&1 | tail -1 >> $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 2 2>/dev/null; }
echo "c5  $(b c5 4)" >> $O/tests.txt
cat $O/tests.txt
