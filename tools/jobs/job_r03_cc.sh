cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_cc
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for pc in 2 1 3 2 1; do
echo "c2 wg bwd workgroups per CU $pc  $(GCPNET_WG_BWD_PER_CU=$pc b c2 20)" >> $O/step.txt
done
for pc in 2 1 2 1; do
echo "c5 wg bwd workgroups per CU $pc  $(GCPNET_WG_BWD_PER_CU=$pc b c5 4)" >> $O/step.txt
done
for pc in 2 1 2 1; do
echo "c3 wg bwd workgroups per CU $pc  $(GCPNET_WG_BWD_PER_CU=$pc b c3 10)" >> $O/step.txt
done
cat $O/step.txt
