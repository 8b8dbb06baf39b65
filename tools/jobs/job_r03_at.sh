cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_at
mkdir -p $O
L=$GRAFT_REPO_ROOT/tools/repro/libgcpnet_hip_tn_rk16.so
GCPNET_HIP_LIB=$L timeout 900 python -m pytest tests/test_tn_gemm.py -q -x -k "default or fp32 or blocked" 2>&1 | tail -3 > $O/tests.txt
GCPNET_HIP_LIB=$L python tools/tn_bench.py 2>/dev/null | grep bf16 > $O/tn.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 2 2>/dev/null; }
echo "c2 rk32  $(b c2 20)" >> $O/step.txt
echo "c2 rk16  $(GCPNET_HIP_LIB=$L b c2 20)" >> $O/step.txt
echo "c2 rk32  $(b c2 20)" >> $O/step.txt
echo "c2 rk16  $(GCPNET_HIP_LIB=$L b c2 20)" >> $O/step.txt
echo "c5 rk32  $(b c5 4)" >> $O/step.txt
echo "c5 rk16  $(GCPNET_HIP_LIB=$L b c5 4)" >> $O/step.txt
cat $O/tests.txt $O/tn.txt $O/step.txt
