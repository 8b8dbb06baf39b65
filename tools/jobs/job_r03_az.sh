cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_az
mkdir -p $O
export PYTHONUNBUFFERED=1
for seed in 21 22 23; do
  SWEEP_POISON=1 SWEEP_REPEAT=2 timeout 1200 python tests/sweep_layers.py 60 $seed > $O/layers_$seed.txt 2>&1
  tail -n 1 $O/layers_$seed.txt
done
for seed in 31 32; do
  SWEEP_POISON=1 SWEEP_REPEAT=2 timeout 1200 python tests/sweep_gcp2.py 150 $seed > $O/gcp2_$seed.txt 2>&1
  tail -n 1 $O/gcp2_$seed.txt
done
grep -c "dims=(36, 8)\|dims=(52, 12)\|dims=(120, 16)\|dims=(100, 16)" $O/layers_*.txt
