cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_cd
mkdir -p $O
GCPNET_TN_SINGLE_BUFFER=1 timeout 900 python -m pytest tests/test_tn_gemm.py -q -x -k "default or blocked" 2>&1 | tail -2 > $O/step.txt
GCPNET_TN_SINGLE_BUFFER=1 python tools/tn_bench.py 2>/dev/null | grep bf16 >> $O/step.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for i in 1 2; do
echo "c2 two buffers  $(b c2 20)" >> $O/step.txt
echo "c2 one buffer   $(GCPNET_TN_SINGLE_BUFFER=1 b c2 20)" >> $O/step.txt
done
echo "c2 one buffer, 256 splits   $(GCPNET_TN_SPLITS=256 GCPNET_TN_SINGLE_BUFFER=1 b c2 20)" >> $O/step.txt
for i in 1 2; do
echo "c5 two buffers  $(b c5 4)" >> $O/step.txt
echo "c5 one buffer   $(GCPNET_TN_SINGLE_BUFFER=1 b c5 4)" >> $O/step.txt
done
cat $O/step.txt
