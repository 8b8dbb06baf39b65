# round 4, job d: how much of the chain backward is the vector-memory path -- tile-blocked / lane-contiguous accesses for s_pre, ds_pre,
# the d(V) state and the norms (EXP3, EXP5), weight fragments from LDS (EXP2), ds_pre behind step E (ORDER1); measurement builds
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_d
mkdir -p $O
for v in FINE3 EXP3EXP5 EXP3EXP5EXP2 EXP3EXP5EXP2ORDER1; do
  echo "== $v" >> $O/phase.txt
  GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python tools/phase_timing.py 160000 128 16 2>&1 | tail -8 >> $O/phase.txt
  echo "$v $(GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python bench.py --no-cpu-baseline --no-c5-block --no-other-configs 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print(d["ms_per_step"], d["roofline"]["all_kernels_ms"])')" >> $O/kern.txt
done
cat $O/phase.txt $O/kern.txt
