cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_u
mkdir -p $O
export PYTHONUNBUFFERED=1
for cfg in c5 c2; do
for mode in "" "--halo"; do
  BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --shard graph $mode --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-c5-block --no-other-configs > $O/bench_${cfg}_shard2$mode.json 2> $O/bench_${cfg}_shard2$mode.err
  python - $O/bench_${cfg}_shard2$mode.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("parallelism"))
except Exception as ex:
    print("FAILED", sys.argv[1], ex)
PY
  grep -A8 Traceback $O/bench_${cfg}_shard2$mode.err | head -20
done; done
