cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bu
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_wg_kernels.py tests/test_trainer_glue.py tests/test_full_size.py tests/test_cpd_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/tests.txt
SWEEP_POISON=1 timeout 900 python tests/sweep_layers.py 60 81 2>&1 | tail -1 >> $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c5  $(b c5 4)" >> $O/tests.txt
echo "c2  $(b c2 20)" >> $O/tests.txt
cat $O/tests.txt
