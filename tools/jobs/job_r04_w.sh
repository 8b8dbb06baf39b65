#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_w; mkdir -p $O
cd $R
timeout 600 python tools/host_enqueue_time.py c2 30 > $O/host_enqueue.txt 2>$O/err.txt
timeout 600 python tools/host_enqueue_time.py c2 30 --nodes 200 >> $O/host_enqueue.txt 2>>$O/err.txt
GCPNET_SIDE_STREAM=0 timeout 600 python tools/host_enqueue_time.py c2 30 >> $O/host_enqueue.txt 2>>$O/err.txt
cat $O/host_enqueue.txt
