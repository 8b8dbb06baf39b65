cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_br
mkdir -p $O
export PYTHONUNBUFFERED=1
for seed in 61 62 63 64; do
  SWEEP_POISON=1 SWEEP_REPEAT=2 timeout 1500 python tests/sweep_layers.py 120 $seed > $O/layers_$seed.txt 2>&1
  echo "layers $seed: $(tail -n 1 $O/layers_$seed.txt)"
done
for seed in 71 72 73; do
  SWEEP_POISON=1 SWEEP_REPEAT=2 timeout 1500 python tests/sweep_gcp2.py 300 $seed > $O/gcp2_$seed.txt 2>&1
  echo "gcp2 $seed: $(tail -n 1 $O/gcp2_$seed.txt)"
done
