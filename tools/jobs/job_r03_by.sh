cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_by
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for sp in 256 128 192 320 384 256; do
echo "c2 splits $sp  $(GCPNET_TN_SPLITS=$sp b c2 20)" >> $O/step.txt
done
for sp in 256 128 384 256; do
echo "c5 splits $sp  $(GCPNET_TN_SPLITS=$sp b c5 4)" >> $O/step.txt
done
cat $O/step.txt
