cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_m
mkdir -p $O
for sp in 64 128 192 256 384 512; do
  echo "== GCPNET_TN_SPLITS=$sp" >> $O/tn_splits.txt
  GCPNET_TN_SPLITS=$sp python tools/tn_bench.py 159913 128 144 999995 256 284 2>/dev/null >> $O/tn_splits.txt
  echo "c2 step $(GCPNET_TN_SPLITS=$sp python bench.py --step-only --steps 20 --warmup 5 2>/dev/null)" >> $O/tn_splits.txt
done
cat $O/tn_splits.txt
