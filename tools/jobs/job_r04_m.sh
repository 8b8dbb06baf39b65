# round 4, job m: batched glue (copy2d_multi, axpy_clamp backward kernel, chain packs in one launch): tests + c1 / c4 / c2 steps
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_m
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_wg_kernels.py tests/test_trainer_glue.py tests/test_side_stream.py tests/test_gcpnet_equivariance.py -m gpu -x -q 2>&1 | tail -4 > $O/tests.txt
for c in c1 c4; do echo "$c graph $(python bench.py --config $c --hip-graph --step-only --steps 30 --warmup 3 2>/dev/null)" >> $O/step.txt; done
echo "c2 $(python bench.py --config c2 --step-only --steps 20 --warmup 5 2>/dev/null)" >> $O/step.txt
cat $O/tests.txt $O/step.txt
