cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_c
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_cpd_gpu.py tests/test_gpu_parity.py -m gpu -q -x -k "cpd or masked" 2>&1 | tail -30 > $O/tests_cpd.txt
python bench.py --step-only --steps 20 --warmup 5 > $O/c2_base.json 2>/dev/null
GCPNET_WG_BWD_NOFUSE=1 python bench.py --step-only --steps 20 --warmup 5 > $O/c2_nofuse.json 2>/dev/null
GCPNET_SIDE_STREAM=0 python bench.py --step-only --steps 20 --warmup 5 > $O/c2_noside.json 2>/dev/null
python bench.py --config c5 --step-only --steps 3 --warmup 2 > $O/c5_base.json 2>/dev/null
GCPNET_SIDE_STREAM=0 python bench.py --config c5 --step-only --steps 3 --warmup 2 > $O/c5_noside.json 2>/dev/null
for c in c1 c3 c4; do python bench.py --config $c --step-only --steps 20 --warmup 5 > $O/${c}_eager.json 2>/dev/null; python bench.py --config $c --hip-graph --step-only --steps 20 --warmup 5 > $O/${c}_graph.json 2>/dev/null; done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --step-only --steps 3 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 6 > $O/c2_timeline.txt 2>&1
find $O/kt -name "*kernel_trace.csv" -size +30M -delete
tail -n 3 $O/*.json $O/tests_cpd.txt
