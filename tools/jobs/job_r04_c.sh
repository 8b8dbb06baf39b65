# round 4, job c: ds_pre stored behind step E (ORDER=1), alone and with the tile-blocked accesses (EXP3)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_c
mkdir -p $O
for v in FINE3 ORDER1 ORDER1EXP3; do
  echo "== $v" >> $O/phase.txt
  GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python tools/phase_timing.py 160000 128 16 2>&1 | tail -10 >> $O/phase.txt
done
GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_ORDER1.so timeout 600 python -m pytest tests/test_wg_kernels.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 > $O/tests_order1.txt
for v in FINE3 ORDER1; do
  echo "$v $(GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python bench.py --step-only --steps 20 --warmup 5 2>/dev/null)" >> $O/step.txt
done
cat $O/phase.txt $O/tests_order1.txt $O/step.txt
