cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_as
mkdir -p $O
timeout 900 python -m pytest tests/test_tn_gemm.py -q -x 2>&1 | tail -3 > $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 2 2>/dev/null; }
echo "c5 default (128x160) $(b c5 4)" >> $O/step.txt
echo "c5 big block         $(GCPNET_TN_BIG=1 b c5 4)" >> $O/step.txt
echo "c5 default (128x160) $(b c5 4)" >> $O/step.txt
cat $O/tests.txt $O/step.txt
