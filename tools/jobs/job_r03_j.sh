cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_j
mkdir -p $O
export TMPDIR=/tmp
python tools/tn_bench.py > $O/tn_bench_big.txt 2>&1
GCPNET_TN_NO_BIG=1 python tools/tn_bench.py > $O/tn_bench_small.txt 2>&1
cd /tmp
for mode in big small; do
  if [ $mode = small ]; then export GCPNET_TN_NO_BIG=1; else unset GCPNET_TN_NO_BIG; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_$mode -- python $GRAFT_REPO_ROOT/tools/tn_bench.py 999995 256 284 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2_$mode -- python $GRAFT_REPO_ROOT/tools/tn_bench.py 999995 256 284 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("O", "gpurun_out/r03_j")
for d in sorted(glob.glob(O + "/pmc*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        print("==", d)
        for k, c in agg.items():
            if "tn_gemm" in k:
                print(k, {n: round(v / cnt[(k, n)]) for n, v in c.items()})
PY
cat $O/tn_bench_big.txt $O/tn_bench_small.txt
find $O -name "*.csv" -size +5M -delete
