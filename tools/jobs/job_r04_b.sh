# round 4, job b: upper bounds before the redesign of the chain backward -- weight fragments from L1 / LDS (EXP1 / EXP2: wrong results),
# tile-blocked s_pre / ds_pre accesses (EXP3: a layout the consumers do not understand); stamps as in the FINE3 build
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_b
mkdir -p $O
for v in FINE3 EXP1 EXP2 EXP3 EXP1EXP3 EXP2EXP3; do
  echo "== $v" >> $O/phase.txt
  GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python tools/phase_timing.py 160000 128 16 2>&1 | tail -10 >> $O/phase.txt
done
cat $O/phase.txt
