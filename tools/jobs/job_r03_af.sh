cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_af
mkdir -p $O
export GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/repro/libgcpnet_hip_tn_timing.so
echo "--- no DMA" > $O/phases.txt
GCPNET_TN_DEBUG=2 python tools/tn_phase_timing.py 159913 128 144 >> $O/phases.txt 2>&1
echo "--- no DMA, no K2" >> $O/phases.txt
GCPNET_TN_NO_K2=1 GCPNET_TN_DEBUG=2 python tools/tn_phase_timing.py 159913 128 144 >> $O/phases.txt 2>&1
echo "--- DMA, no K2" >> $O/phases.txt
GCPNET_TN_NO_K2=1 python tools/tn_phase_timing.py 159913 128 144 >> $O/phases.txt 2>&1
cat $O/phases.txt
