cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ah
mkdir -p $O
timeout 900 python -m pytest tests/test_tn_gemm.py -q -x 2>&1 | tail -8 > $O/tests.txt
GCPNET_TN_FOUR_WAVES=1 timeout 900 python -m pytest tests/test_tn_gemm.py -q -x 2>&1 | tail -3 >> $O/tests.txt
python tools/tn_bench.py 2>/dev/null > $O/tn.txt
echo "four waves" >> $O/tn.txt
GCPNET_TN_FOUR_WAVES=1 python tools/tn_bench.py 2>/dev/null | grep bf16 >> $O/tn.txt
GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/repro/libgcpnet_hip_tn_timing.so python tools/tn_phase_timing.py 159913 128 144 > $O/phases.txt 2>&1
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
echo "c2 x3 8w    $(b c2 20)" >> $O/step.txt
echo "c2 x3 4w    $(GCPNET_TN_FOUR_WAVES=1 b c2 20)" >> $O/step.txt
echo "c2 fp32     $(GCPNET_TN_FP32=1 b c2 20)" >> $O/step.txt
echo "c2 x3 8w    $(b c2 20)" >> $O/step.txt
echo "c5 x3 8w    $(b c5 4)" >> $O/step.txt
echo "c5 fp32     $(GCPNET_TN_FP32=1 b c5 4)" >> $O/step.txt
cat $O/tests.txt $O/tn.txt $O/phases.txt $O/step.txt
