cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_d
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_cpd_gpu.py -m gpu -q -x 2>&1 | tail -30 > $O/tests_cpd.txt
timeout 300 python tools/diag_wg_support.py 256 32 3200 > $O/diag_c5_ff.txt 2>&1
tail -n 40 $O/tests_cpd.txt $O/diag_c5_ff.txt
