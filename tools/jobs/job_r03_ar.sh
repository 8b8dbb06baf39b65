cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ar
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 2 2>/dev/null; }
echo "c5 default          $(b c5 4)" >> $O/step.txt
echo "c5 side stream off  $(GCPNET_SIDE_STREAM=0 b c5 4)" >> $O/step.txt
echo "c5 no big TN        $(GCPNET_TN_NO_BIG=1 b c5 4)" >> $O/step.txt
echo "c5 TN 8 waves       $(GCPNET_TN_EIGHT_WAVES=1 b c5 4)" >> $O/step.txt
echo "c5 default          $(b c5 4)" >> $O/step.txt
echo "c2 side stream off  $(GCPNET_SIDE_STREAM=0 b c2 20)" >> $O/step.txt
echo "c2 default          $(b c2 20)" >> $O/step.txt
cat $O/step.txt
