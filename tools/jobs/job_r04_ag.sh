#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for cfg in c3 c2 c5; do
  st=30; [ $cfg = c5 ] && st=6
  timeout 600 python bench.py --config $cfg --step-only --steps $st --warmup 5 2>>$O/err.txt | tail -1
done
