#!/bin/bash
# GPU box: times the (256,32) block backward of every tools/variants/libgcpnet_hip_wgb_*.so (tools/wg_block_probe.py)
R=$PWD; OUT=$R/gpurun_out/${1:-wgx}.txt; : > $OUT
for L in $(ls $R/tools/variants/libgcpnet_hip_wgb_*.so | sort -V); do
  echo "== $(basename $L)  $(GCPNET_HIP_LIB=$L python $R/tools/wg_block_probe.py 2>&1 | grep bwd_ms)" >> $OUT
done
cat $OUT
