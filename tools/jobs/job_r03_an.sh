cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_an
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_wg_kernels.py tests/test_bf16x3.py -m gpu -q -x -k "chain" 2>&1 | tail -8 > $O/tests.txt
timeout 900 python tests/sweep_layers.py 40 7 2>&1 | tail -5 >> $O/tests.txt
cat $O/tests.txt
