cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_k
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/test_full_size.py -m gpu -q -x -s 2>&1 | tail -40 > $O/tests_full.txt
SWEEP_POISON=1 SWEEP_REPEAT=2 timeout 900 python -u tests/sweep_layers.py 80 1 2>&1 | grep -n " 68 N\|mismatches" > $O/sweep_layers_1.txt
tail -n 30 $O/tests_full.txt; cat $O/sweep_layers_1.txt
