#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_y; mkdir -p $O
cd $R
timeout 600 python tools/find_aten_ops.py c1 > $O/aten_c1.txt 2>$O/err.txt
head -30 $O/aten_c1.txt
