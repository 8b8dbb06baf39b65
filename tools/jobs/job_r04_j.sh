# round 4, job j: how much of the (256,32) block backward is its weight stream -- P4 with every fragment from 6 KB (L1-resident): EXP1
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_j
mkdir -p $O
for v in base EXP1; do
  L=""; [ $v != base ] && L=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_wgb_$v.so
  echo "== $v" >> $O/phase.txt
  GCPNET_HIP_LIB=$L python tools/wg_phase_timing.py 500000 256 32 2>&1 | grep -v amdgpu | tail -16 >> $O/phase.txt
done
cat $O/phase.txt
