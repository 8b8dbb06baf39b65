cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_f
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_wg_kernels.py tests/test_sharded_gpu.py tests/test_rccl_world1.py -m gpu -q -x 2>&1 | tail -30 > $O/tests.txt
GCPNET_DEBUG_UNSUPPORTED=1 timeout 300 python tools/diag_wg_support.py 256 32 3200 > $O/diag_c5_ff.txt 2>&1
python bench.py --config c5 --step-only --steps 3 --warmup 2 > $O/c5.json 2>/dev/null
python bench.py --config c5 --dry-run-world 8 > $O/c5_dry8.json 2>$O/c5_dry8.err
python bench.py --config c2 --dry-run-world 8 > $O/c2_dry8.json 2>>$O/c5_dry8.err
tail -n 25 $O/tests.txt $O/diag_c5_ff.txt $O/c5.json; head -c 1500 $O/c5_dry8.json; tail -3 $O/c5_dry8.err
