cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_l
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { python bench.py --step-only --steps 20 --warmup 5 "$@" 2>/dev/null; }
echo "c2 base      $(b)" > $O/knobs.txt
echo "c2 defer tn  $(GCPNET_DEFER_TN=1 b)" >> $O/knobs.txt
echo "c2 base      $(b)" >> $O/knobs.txt
echo "c2 defer tn  $(GCPNET_DEFER_TN=1 b)" >> $O/knobs.txt
echo "c5 base      $(python bench.py --config c5 --step-only --steps 3 --warmup 2 2>/dev/null)" >> $O/knobs.txt
echo "c5 defer tn  $(GCPNET_DEFER_TN=1 python bench.py --config c5 --step-only --steps 3 --warmup 2 2>/dev/null)" >> $O/knobs.txt
for c in c1 c4; do echo "$c eager    $(b --config $c)" >> $O/knobs.txt; done
timeout 600 python -m pytest tests/test_side_stream.py -m gpu -q -x 2>&1 | tail -3 >> $O/knobs.txt
GCPNET_DEFER_TN=1 timeout 600 python -m pytest tests/test_side_stream.py tests/test_full_size.py -m gpu -q -x -k "c2 or side" 2>&1 | tail -3 >> $O/knobs.txt
export BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 5 --warmup 2 > $O/gpus2_batch.json 2> $O/gpus2_batch.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --steps 5 --warmup 2 --shard graph > $O/gpus2_graph.json 2> $O/gpus2_graph.err
cat $O/knobs.txt; head -c 600 $O/gpus2_batch.json; echo; head -c 600 $O/gpus2_graph.json; echo; tail -3 $O/gpus2_batch.err $O/gpus2_graph.err
