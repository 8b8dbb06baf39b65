cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_x
mkdir -p $O
timeout 900 python -m pytest tests/test_tn_gemm.py -q -x 2>&1 | tail -15 > $O/tests.txt
for sp in 256 512 768; do
  echo "splits $sp" >> $O/tn.txt
  GCPNET_TN_SPLITS=$sp python tools/tn_bench.py 2>/dev/null >> $O/tn.txt
done
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c2 x3       $(b c2 20)" >> $O/step.txt
echo "c2 fp32     $(GCPNET_TN_FP32=1 b c2 20)" >> $O/step.txt
echo "c2 x3 sp512 $(GCPNET_TN_SPLITS=512 b c2 20)" >> $O/step.txt
echo "c5 x3       $(b c5 4)" >> $O/step.txt
echo "c5 fp32     $(GCPNET_TN_FP32=1 b c5 4)" >> $O/step.txt
cat $O/tests.txt $O/tn.txt $O/step.txt
