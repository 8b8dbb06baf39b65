cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_g
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/tests_all.txt
GCPNET_DEBUG_UNSUPPORTED=1 timeout 300 python tools/diag_wg_support.py 256 32 3200 > $O/diag_c5_ff.txt 2>&1
python bench.py --config c5 --step-only --steps 3 --warmup 2 > $O/c5.json 2>/dev/null
python bench.py --step-only --steps 20 --warmup 5 > $O/c2.json 2>/dev/null
tail -n 12 $O/tests_all.txt $O/diag_c5_ff.txt; cat $O/c5.json $O/c2.json
