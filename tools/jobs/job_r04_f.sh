# round 4, job f: tile-blocked s_pre / ds_pre / chain states: GPU tests of the chain paths, bench A/B by GCPNET_CHAIN_TB
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_f
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/tests.txt
for tb in 1 0; do
  echo "TB=$tb $(GCPNET_CHAIN_TB=$tb python bench.py --no-cpu-baseline --no-c5-block --no-other-configs 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print(d["ms_per_step"], d["ms_per_step_median"], d["roofline"]["all_kernels_ms"])')" >> $O/kern.txt
done
python bench.py --no-c5-block --no-other-configs > $O/bench_c2.json 2> $O/bench_c2.err
cat $O/tests.txt $O/kern.txt; tail -c 900 $O/bench_c2.json; tail -3 $O/bench_c2.err
