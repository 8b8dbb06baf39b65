# round 4, job e: chain backward without stores in step A (norms through LDS, small stores behind the s_pre requests): V4
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_e
mkdir -p $O
for v in FINE3 $VARIANTS; do
  echo "== $v" >> $O/phase.txt
  GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python tools/phase_timing.py 160000 128 16 2>&1 | tail -8 >> $O/phase.txt
  GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so timeout 900 python -m pytest tests/test_wg_kernels.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 >> $O/phase.txt
  echo "$v $(GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python bench.py --no-cpu-baseline --no-c5-block --no-other-configs 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print(d["ms_per_step"], d["roofline"]["all_kernels_ms"])')" >> $O/kern.txt
done
cat $O/phase.txt $O/kern.txt
