# round 4, job p: planes kernel for the wide problems only; the layer's weight-gradient jobs as bench.py times them
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p
mkdir -p $O
timeout 600 python -m pytest tests/test_tn_gemm.py tests/test_wg_kernels.py -m gpu -x -q 2>&1 | tail -2 > $O/tests.txt
for pl in 0 1; do
  echo "planes=$pl $(GCPNET_TN_PLANES=$pl python bench.py --no-cpu-baseline --no-c5-block --no-other-configs 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print(d["ms_per_step"], d["roofline"]["all_kernels_ms"])')" >> $O/kern.txt
done
for pl in 0 1; do echo "c5 planes=$pl $(GCPNET_TN_PLANES=$pl python bench.py --config c5 --step-only --steps 4 --warmup 3 2>/dev/null)" >> $O/kern.txt; done
cat $O/tests.txt $O/kern.txt
