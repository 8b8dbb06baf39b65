cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bq
mkdir -p $O
export PYTHONUNBUFFERED=1
run() { BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-c5-block --no-other-configs; }
run > $O/replicas.json 2> $O/replicas.err
run --shard graph > $O/shard.json 2> $O/shard.err
run --shard graph --halo > $O/halo.json 2> $O/halo.err
python - <<'PY'
import json
for n in ("replicas", "shard", "halo"):
    try:
        d = json.loads(open(f"gpurun_out/r03_bq/{n}.json").read().strip().splitlines()[-1])
        print(n, d["n_gpus"], d["ms_per_step"], d["value"], d["scaling"], d["config"].get("parallelism", "")[:90])
    except Exception as ex:
        print(n, "FAILED", ex)
PY
