cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_v
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python tools/halo_exchange_timing.py 10000 128 2>&1 | grep halo= > $O/timing.txt
timeout 600 python tools/halo_exchange_timing.py 100000 256 2>&1 | grep halo= >> $O/timing.txt
timeout 1200 python -m pytest tests/test_sharded_gpu.py -q -x 2>&1 | tail -3 >> $O/timing.txt
for mode in "" "--halo"; do
  BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --shard graph $mode --config c2 --steps 3 --warmup 2 --no-cpu-baseline --no-c5-block --no-other-configs > $O/bench_c2_shard2$mode.json 2> $O/bench_c2_shard2$mode.err
  python - $O/bench_c2_shard2$mode.json <<'PY' >> $O/timing.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"])
except Exception as ex:
    print("FAILED", sys.argv[1], ex)
PY
done
cat $O/timing.txt
