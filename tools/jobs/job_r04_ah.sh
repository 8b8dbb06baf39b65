#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ah; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_wg_kernels.py -x -q -m gpu -k "tile_blocked_route or message_chain" > $O/tests1.txt 2>&1
tail -15 $O/tests1.txt
for tb in 1 0; do
  GCPNET_CHAIN_TB=$tb timeout 600 python bench.py --config c5 --step-only --steps 6 --warmup 3 2>>$O/err.txt | tail -1
done
