cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_s
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_wg_kernels.py tests/test_gpu_parity.py tests/test_bf16x3.py tests/test_gcpnet_equivariance.py -m gpu -q -x 2>&1 | tail -4 > $O/tests.txt
python bench.py --no-cpu-baseline --no-c5-block --no-other-configs > $O/bench_short.json 2>/dev/null
python - <<'PY' >> $O/tests.txt
import json
d = json.load(open("gpurun_out/r03_s/bench_short.json"))
print(d["ms_per_step"], d["ms_per_step_median"], d["roofline"]["all_kernels_ms"])
PY
cat $O/tests.txt
