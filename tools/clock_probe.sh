#!/bin/bash
# What clock / power does the box run the chain kernels at?  Samples rocm-smi while a loop of chain launches runs, then lists the
# instruction-cache counters rocprofv3 offers.  usage (GPU box, repo root): tools/clock_probe.sh <tag>
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/${1:-clk}; mkdir -p $OUT
rocm-smi --showclocks --showpower --showperflevel > $OUT/smi_idle.txt 2>&1
rocm-smi --showmaxpower --showclkfrq > $OUT/smi_caps.txt 2>&1
python $R/tools/chain_rows_sweep.py --tiles 4998,4998,4998,4998,4998,4998,4998,4998 --iters 300 > $OUT/loop.txt 2>&1 &
PID=$!
sleep 8
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" >> $OUT/smi_busy.txt
  echo "--" >> $OUT/smi_busy.txt
  sleep 0.5
done
wait $PID
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|ifetch\|SQ_WAIT\|SQ_INST_CYCLES\|SQ_INSTS_\|SQC_" | head -80 > $OUT/counters.txt
