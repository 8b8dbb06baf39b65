cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_aw
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/test_wg_kernels.py tests/test_gpu_parity.py tests/test_side_stream.py tests/test_rccl_world1.py tests/test_trainer_glue.py tests/test_cpd_gpu.py tests/test_sweeps_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/tests.txt
python bench.py --no-cpu-baseline --no-c5-block > $O/bench.json 2>/dev/null
python - <<'PY' >> $O/tests.txt
import json
d = json.loads(open("gpurun_out/r03_aw/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], {k: (v["eager_ms_per_step_median"], v["hipgraph_ms_per_step_median"]) for k, v in d["other_configs"].items()})
PY
cat $O/tests.txt
