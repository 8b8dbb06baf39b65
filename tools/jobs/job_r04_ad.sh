#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ad; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for cfg in c1 c4 c3; do
  for b in 1 0; do
    echo "== $cfg batch_packs=$b" >> $O/ab.txt
    GCPNET_BATCH_WG_PACKS=$b timeout 600 python bench.py --config $cfg --step-only --steps 30 --warmup 5 --hip-graph 2>>$O/err.txt | tail -1 >> $O/ab.txt
    GCPNET_BATCH_WG_PACKS=$b timeout 600 python bench.py --config $cfg --step-only --steps 30 --warmup 5 2>>$O/err.txt | tail -1 >> $O/ab.txt
  done
done
cat $O/ab.txt
