cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_av
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 3000 python -m pytest tests -m gpu -q --durations=10 2>&1 | tail -30 > $O/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -n 22 $O/tests_all.txt; tail -n 2 $O/smoke.txt
