#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v; mkdir -p $O
cd $R
HOST_PROFILE_BODIES=1 timeout 600 python tools/host_backward_profile.py c2 20 --nodes 200 > $O/host_c2_tiny_bodies.txt 2>$O/err.txt
head -5 $O/host_c2_tiny_bodies.txt
