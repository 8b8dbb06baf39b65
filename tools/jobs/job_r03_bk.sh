cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bk
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for us in 0 100 200 400 0; do
echo "c2 spin $us us before each chain backward   $(GCPNET_DEBUG_SPIN_US=$us b c2 20)" >> $O/step.txt
done
cat $O/step.txt
