cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_aj
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for f in 1 0.9 0.8 0.7 0.6 0.5 1; do
echo "c2 tail-inline $f   $(GCPNET_SIDE_TAIL_INLINE=$f b c2 20)" >> $O/step.txt
done
for f in 1 0.8 0.6; do
echo "c5 tail-inline $f   $(GCPNET_SIDE_TAIL_INLINE=$f b c5 4)" >> $O/step.txt
done
cat $O/step.txt
