cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_p
mkdir -p $O
c5() { python bench.py --config c5 --step-only --steps 3 --warmup 2 2>/dev/null; }
for k in 96 300 2000; do echo "head4_k $k  $(GCPNET_WG_FWD_HEAD4_K=$k c5)" >> $O/c5.txt; done
echo "head4_k 96   $(GCPNET_WG_FWD_HEAD4_K=96 c5)" >> $O/c5.txt
cat $O/c5.txt
