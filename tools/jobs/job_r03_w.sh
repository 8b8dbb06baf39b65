cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_w
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c2 base     $(b c2 20)" >> $O/skip.txt
echo "c2 skip TN  $(GCPNET_DEBUG_SKIP_TN=1 b c2 20)" >> $O/skip.txt
echo "c2 base     $(b c2 20)" >> $O/skip.txt
echo "c2 skip TN  $(GCPNET_DEBUG_SKIP_TN=1 b c2 20)" >> $O/skip.txt
echo "c5 base     $(b c5 4)" >> $O/skip.txt
echo "c5 skip TN  $(GCPNET_DEBUG_SKIP_TN=1 b c5 4)" >> $O/skip.txt
cat $O/skip.txt
