#!/bin/bash
# c2 / c5 step: eager against hipGraph replay, and the launch timeline of one eager c2 step on both queues
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_t; mkdir -p $O
cd $R
for mode in "" "--hip-graph"; do
  echo "== c2 $mode" >> $O/ab.txt
  timeout 600 python bench.py --config c2 --step-only --steps 30 --warmup 5 $mode 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_median'), d.get('config'))" >> $O/ab.txt
done
for mode in "" "--hip-graph"; do
  echo "== c5 $mode" >> $O/ab.txt
  timeout 900 python bench.py --config c5 --step-only --steps 6 --warmup 2 $mode 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_median'), d.get('config'))" >> $O/ab.txt
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ks -- python $R/bench.py --config c2 --step-only --steps 6 --warmup 3 > $O/prof_bench.txt 2>>$O/err.txt
f=$(find $O/ks -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f 9 0 gcp2_chain_fwd_kernel:4 > $O/c2_timeline.txt 2>>$O/err.txt
rm -rf $O/ks
tail -5 $O/ab.txt
