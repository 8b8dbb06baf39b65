cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ab
mkdir -p $O
GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/repro/libgcpnet_hip_tn_timing.so python tools/tn_phase_timing.py > $O/phases.txt 2>&1
echo "--- no DMA" >> $O/phases.txt
GCPNET_TN_DEBUG=2 GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/repro/libgcpnet_hip_tn_timing.so python tools/tn_phase_timing.py >> $O/phases.txt 2>&1
cat $O/phases.txt
