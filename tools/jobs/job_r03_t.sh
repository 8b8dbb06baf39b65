cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_t
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_sharded_gpu.py tests/test_rccl_world1.py tests/test_parallel_gloo.py -q -x 2>&1 | tail -6 > $O/tests.txt
# two ranks on the one GPU over gloo: the sharded bench in both exchange modes (rehearsal of the driver's multi-GPU launch)
for mode in "" "--halo"; do
  BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --shard graph $mode --config c5 --steps 3 --warmup 2 --no-cpu-baseline --no-c5-block --no-other-configs > $O/bench_shard2$mode.json 2> $O/bench_shard2$mode.err
  tail -c 1500 $O/bench_shard2$mode.json; tail -n 3 $O/bench_shard2$mode.err
done
python bench.py --config c5 --dry-run-world 8 > $O/c5_dry8.json 2>/dev/null
python bench.py --config c2 --dry-run-world 8 > $O/c2_dry8.json 2>/dev/null
cat $O/tests.txt; tail -c 1200 $O/c5_dry8.json
