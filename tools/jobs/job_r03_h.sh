cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_h
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_tn_gemm.py tests/test_sharded_gpu.py tests/test_wg_kernels.py tests/test_full_size.py -m gpu -q 2>&1 | tail -40 > $O/tests.txt
python bench.py --config c5 --step-only --steps 3 --warmup 2 > $O/c5_big.json 2>/dev/null
GCPNET_TN_NO_BIG=1 python bench.py --config c5 --step-only --steps 3 --warmup 2 > $O/c5_nobig.json 2>/dev/null
tail -n 15 $O/tests.txt; cat $O/c5_big.json $O/c5_nobig.json
