#!/bin/bash
# hipGraph replay with the weight-gradient stream kept as a parallel branch of the graph
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_x; mkdir -p $O
cd $R
for cfg in c2 c1 c4 c3 c5; do
  st=30; [ $cfg = c5 ] && st=6
  echo "== $cfg eager" >> $O/ab.txt
  timeout 600 python bench.py --config $cfg --step-only --steps $st --warmup 5 2>>$O/err.txt | tail -1 >> $O/ab.txt
  for ss in 0 1; do
    echo "== $cfg hip-graph side=$ss" >> $O/ab.txt
    GCPNET_GRAPH_SIDE_STREAM=$ss timeout 600 python bench.py --config $cfg --step-only --steps $st --warmup 5 --hip-graph 2>>$O/err.txt | tail -1 >> $O/ab.txt
  done
done
cat $O/ab.txt
