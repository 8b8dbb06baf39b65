#!/bin/bash
# host side of the c2 step: step time on a tiny graph (same launches, no GPU work = the host's time per step), cProfile of the full step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_u; mkdir -p $O
cd $R
echo "== c2 with 200 nodes (host-bound)" > $O/host.txt
timeout 600 python bench.py --config c2 --nodes 200 --step-only --steps 30 --warmup 5 2>>$O/err.txt | tail -1 >> $O/host.txt
echo "== c2 full" >> $O/host.txt
timeout 600 python bench.py --config c2 --step-only --steps 30 --warmup 5 2>>$O/err.txt | tail -1 >> $O/host.txt
timeout 600 python tools/pyprofile_step.py c2 20 > $O/pyprof_c2.txt 2>>$O/err.txt
timeout 600 python tools/pyprofile_step.py c2 20 --nodes 200 > $O/pyprof_c2_tiny.txt 2>>$O/err.txt
cat $O/host.txt
