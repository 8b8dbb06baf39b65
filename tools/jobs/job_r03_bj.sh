cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bj
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 2 2>/dev/null; }
echo "c5 default          $(b c5 4)" >> $O/step.txt
echo "c5 chain TN inline  $(GCPNET_CHAIN_TN_INLINE=1 b c5 4)" >> $O/step.txt
echo "c5 default          $(b c5 4)" >> $O/step.txt
echo "c5 chain TN inline  $(GCPNET_CHAIN_TN_INLINE=1 b c5 4)" >> $O/step.txt
echo "c5 inline + defer   $(GCPNET_DEFER_TN=1 b c5 4)" >> $O/step.txt
cat $O/step.txt
