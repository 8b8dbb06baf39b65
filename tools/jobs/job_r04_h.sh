# round 4, job h: new tests (neighbour cap, ReLU census on the configs[4] sub-problem, silu element-wise model steps), the flagged
# leakyrelu sweep case with the pre-activations printed
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_h
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests/test_full_size.py tests/test_trainer_glue.py -m gpu -x -q -s 2>&1 | grep -v Warning | tail -25 > $O/tests.txt
SWEEP_ONLY=72 SWEEP_VERBOSE=72 timeout 600 python tests/sweep_layers.py 73 64 2>&1 | grep -v amdgpu | cut -c1-300 > $O/sweep_case72.txt
cat $O/tests.txt; cat $O/sweep_case72.txt
