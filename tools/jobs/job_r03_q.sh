cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_q
mkdir -p $O
c5() { python bench.py --config c5 --step-only --steps 3 --warmup 2 2>/dev/null; }
echo "base        $(c5)" >> $O/c5.txt
echo "bwd head4   $(GCPNET_WG_BWD_HEAD4=1 c5)" >> $O/c5.txt
echo "base        $(c5)" >> $O/c5.txt
echo "bwd head4   $(GCPNET_WG_BWD_HEAD4=1 c5)" >> $O/c5.txt
GCPNET_WG_BWD_HEAD4=1 timeout 600 python -m pytest tests/test_wg_kernels.py -m gpu -q -x 2>&1 | tail -3 >> $O/c5.txt
cat $O/c5.txt
