# round 4, job g: capturable Adam test, bench with the new blocks (c5 kernel roofline, weight-gradient parity, env / rccl echo)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_g
mkdir -p $O
timeout 900 python -m pytest tests/test_trainer_glue.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/tests.txt; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_g/bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["all_kernels_ms"])
print("env", d["env"], "rccl", d["rccl"], "saved/layer", d.get("saved_activation_bytes_per_layer"))
print("parity", d["parity_check"])
for k,v in d["other_configs"].items(): print(k, v.get("eager_ms_per_step_median"), v.get("hipgraph_ms_per_step_median"), v.get("hipgraph_error"))
c5=d["c5_single_gpu"]; print("c5", c5["ms_per_step"], c5.get("saved_activation_bytes_per_layer"), c5.get("roofline"), c5.get("roofline_error"))
PY
tail -3 $O/bench.err
