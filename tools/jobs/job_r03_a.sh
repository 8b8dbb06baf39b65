cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_a
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_side_stream.py tests/test_wg_kernels.py tests/test_full_size.py -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r03_a/tests1.txt
for seed in 0 1 2; do
  SWEEP_POISON=1 SWEEP_REPEAT=3 timeout 900 python -u tests/sweep_gcp2.py 200 $seed > gpurun_out/r03_a/sweep_gcp2_$seed.txt 2>&1
  echo "rc $?" >> gpurun_out/r03_a/sweep_gcp2_$seed.txt
done
for seed in 0 1; do
  SWEEP_POISON=1 SWEEP_REPEAT=3 timeout 900 python -u tests/sweep_layers.py 80 $seed > gpurun_out/r03_a/sweep_layers_$seed.txt 2>&1
  echo "rc $?" >> gpurun_out/r03_a/sweep_layers_$seed.txt
done
tail -3 gpurun_out/r03_a/*.txt
