# round 4, job n: the planes form of the weight-gradient GEMM: tests, stand-alone launch times, steps A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_n
mkdir -p $O
timeout 900 python -m pytest tests/test_tn_gemm.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.txt
for pl in 0 1; do
  echo "== GCPNET_TN_PLANES=$pl" >> $O/tn.txt
  GCPNET_TN_PLANES=$pl python tools/tn_bench.py 2>/dev/null | grep -v amdgpu >> $O/tn.txt
done
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for i in 1 2; do for pl in 0 1; do
echo "c2 planes=$pl $(GCPNET_TN_PLANES=$pl b c2 20)" >> $O/step.txt
done; done
for pl in 0 1; do echo "c5 planes=$pl $(GCPNET_TN_PLANES=$pl b c5 4)" >> $O/step.txt; done
for pl in 0 1; do echo "c3 planes=$pl $(GCPNET_TN_PLANES=$pl b c3 10)" >> $O/step.txt; done
cat $O/tests.txt $O/tn.txt $O/step.txt
