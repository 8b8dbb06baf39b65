cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bx
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for i in 1 2 3; do
echo "c5 split Function  $(b c5 4)" >> $O/step.txt
echo "c5 plain slicing   $(GCPNET_AB_NOSPLITCOLS=1 b c5 4)" >> $O/step.txt
done
cat $O/step.txt
