# round 4, job a: baseline of the round (GPU tests, default bench line, chain-backward phase stamps incl. fine variants)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err
python tools/phase_timing.py 160000 128 16 > $O/phase.txt 2>&1
for v in FINE2 FINE3; do
  GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libgcpnet_hip_cb_$v.so python tools/phase_timing.py 160000 128 16 2>&1 | tail -9 > $O/phase_$v.txt
done
cat $O/tests.txt; tail -c 1500 $O/bench.json; tail -12 $O/phase.txt; cat $O/phase_FINE2.txt $O/phase_FINE3.txt
