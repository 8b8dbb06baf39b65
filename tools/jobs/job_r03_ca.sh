cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ca
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 "${@:3}" 2>/dev/null; }
for i in 1 2 3; do
for sp in 256 128; do
echo "c3 splits $sp  $(GCPNET_TN_SPLITS=$sp b c3 10)" >> $O/step.txt
done; done
for sp in 256 128 256 128; do
echo "c4 graph splits $sp  $(GCPNET_TN_SPLITS=$sp b c4 20 --hip-graph)" >> $O/step.txt
done
for sp in 256 128 256 128; do
echo "c2 splits $sp  $(GCPNET_TN_SPLITS=$sp b c2 20)" >> $O/step.txt
done
cat $O/step.txt
