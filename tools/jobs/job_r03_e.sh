cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_e
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_cpd_gpu.py tests/test_wg_kernels.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -30 > $O/tests.txt
GCPNET_DEBUG_UNSUPPORTED=1 timeout 300 python tools/diag_wg_support.py 256 32 3200 > $O/diag_c5_ff.txt 2>&1
b() { python bench.py --step-only --steps 20 --warmup 5 "$@" 2>/dev/null; }
echo "base        $(b)" > $O/c2_knobs.txt
echo "fn5         $(GCPNET_WG_BWD_FN5=1 b)" >> $O/c2_knobs.txt
for n in 32 64 96 128 192; do echo "cumask $n   $(GCPNET_SIDE_CU_MASK=$n b)" >> $O/c2_knobs.txt; done
for pad in 16384 49152; do echo "tnpad $pad  $(GCPNET_TN_LDS_PAD=$pad b)" >> $O/c2_knobs.txt; done
echo "cumask 128 + pad 16384 $(GCPNET_SIDE_CU_MASK=128 GCPNET_TN_LDS_PAD=16384 b)" >> $O/c2_knobs.txt
c5() { python bench.py --config c5 --step-only --steps 3 --warmup 2 "$@" 2>/dev/null; }
echo "base        $(c5)" > $O/c5_knobs.txt
echo "cumask 64   $(GCPNET_SIDE_CU_MASK=64 c5)" >> $O/c5_knobs.txt
echo "cumask 128  $(GCPNET_SIDE_CU_MASK=128 c5)" >> $O/c5_knobs.txt
echo "tnpad 16384 $(GCPNET_TN_LDS_PAD=16384 c5)" >> $O/c5_knobs.txt
cat $O/c2_knobs.txt $O/c5_knobs.txt; tail -n 25 $O/tests.txt $O/diag_c5_ff.txt
