#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ae; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 600 python tools/host_enqueue_time.py c2 30 > $O/host_enqueue.txt 2>$O/err.txt
timeout 600 python tools/host_backward_profile.py c2 20 --nodes 200 > $O/host_bodies.txt 2>>$O/err.txt
cat $O/host_enqueue.txt; head -10 $O/host_bodies.txt
for i in 1 2 3; do timeout 600 python bench.py --config c2 --step-only --steps 30 --warmup 5 2>>$O/err.txt | tail -1; done
timeout 600 python bench.py --config c3 --step-only --steps 30 --warmup 5 2>>$O/err.txt | tail -1
