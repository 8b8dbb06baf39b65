#!/bin/bash
# chain forward with the small vector weight fragments requested ahead of the stores (PRE) against the build without it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ac; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_wg_kernels.py -x -q -m gpu 2>&1 | tail -2 > $O/tests.txt
for rep in 1 2; do
for v in PRE NOPRE; do
  lib=""; [ $v = NOPRE ] && lib=$R/tools/variants/libgcpnet_hip_cf_nopre.so
  GCPNET_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-c5-block --no-other-configs --steps 30 --warmup 5 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline'].get('all_kernels_ms'))" >> $O/ab.txt
done
done
cat $O/tests.txt $O/ab.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ks -- python $R/bench.py --config c2 --step-only --steps 6 --warmup 3 > $O/prof_bench.txt 2>>$O/err.txt
f=$(find $O/ks -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f 9 0 gcp2_chain_fwd_kernel:4 > $O/c2_timeline.txt 2>>$O/err.txt
rm -rf $O/ks
tail -1 $O/c2_timeline.txt
