cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_aa
mkdir -p $O
for d in 0 1 2 3; do
  echo "debug $d (1 = no products, 2 = no DMA)" >> $O/tn.txt
  GCPNET_TN_DEBUG=$d python tools/tn_bench.py 2>/dev/null | grep bf16 >> $O/tn.txt
done
cat $O/tn.txt
