cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bi
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 2 2>/dev/null; }
echo "c5 default        $(b c5 4)" >> $O/step.txt
echo "c5 TN per block   $(GCPNET_TN_PER_BLOCK=1 b c5 4)" >> $O/step.txt
echo "c5 default        $(b c5 4)" >> $O/step.txt
echo "c5 TN per block   $(GCPNET_TN_PER_BLOCK=1 b c5 4)" >> $O/step.txt
cat $O/step.txt
