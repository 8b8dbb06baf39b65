cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bz
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for sp in 128 64 96 160 128 64 96; do
echo "c2 splits $sp  $(GCPNET_TN_SPLITS=$sp b c2 20)" >> $O/step.txt
done
for sp in 128 64 96 128; do
echo "c5 splits $sp  $(GCPNET_TN_SPLITS=$sp b c5 4)" >> $O/step.txt
done
for sp in 128 64; do
echo "c3 splits $sp  $(GCPNET_TN_SPLITS=$sp b c3 10)" >> $O/step.txt
done
echo "c3 splits 256  $(GCPNET_TN_SPLITS=256 b c3 10)" >> $O/step.txt
cat $O/step.txt
