cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ay
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for i in 1 2; do
echo "c2 batched    $(b c2 20)" >> $O/step.txt
echo "c2 unbatched  $(GCPNET_AB_UNBATCHED=1 b c2 20)" >> $O/step.txt
done
echo "c4 batched    $(b c4 20)" >> $O/step.txt
echo "c4 unbatched  $(GCPNET_AB_UNBATCHED=1 b c4 20)" >> $O/step.txt
echo "c5 batched    $(b c5 4)" >> $O/step.txt
echo "c5 unbatched  $(GCPNET_AB_UNBATCHED=1 b c5 4)" >> $O/step.txt
cat $O/step.txt
