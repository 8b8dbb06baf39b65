cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_y
mkdir -p $O
timeout 900 python -m pytest tests/test_tn_gemm.py -q -x 2>&1 | tail -5 > $O/tests.txt
GCPNET_TN_BLOCKED=1 timeout 900 python -m pytest tests/test_tn_gemm.py -q -x 2>&1 | tail -3 >> $O/tests.txt
for sp in 256 512; do
  echo "cyclic splits $sp" >> $O/tn.txt
  GCPNET_TN_SPLITS=$sp python tools/tn_bench.py 2>/dev/null >> $O/tn.txt
  echo "blocked splits $sp" >> $O/tn.txt
  GCPNET_TN_BLOCKED=1 GCPNET_TN_SPLITS=$sp python tools/tn_bench.py 2>/dev/null | grep bf16 >> $O/tn.txt
done
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c2 cyclic       $(b c2 20)" >> $O/step.txt
echo "c2 blocked      $(GCPNET_TN_BLOCKED=1 b c2 20)" >> $O/step.txt
echo "c2 cyclic sp512 $(GCPNET_TN_SPLITS=512 b c2 20)" >> $O/step.txt
echo "c5 cyclic       $(b c5 4)" >> $O/step.txt
echo "c5 cyclic sp512 $(GCPNET_TN_SPLITS=512 b c5 4)" >> $O/step.txt
cat $O/tests.txt $O/tn.txt $O/step.txt
