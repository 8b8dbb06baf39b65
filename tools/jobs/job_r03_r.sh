cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_r
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 3000 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -45 > $O/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -n 30 $O/tests_all.txt; cat $O/smoke.txt | tail -2
