cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_bs
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_wg_kernels.py tests/test_trainer_glue.py -m gpu -q -x 2>&1 | tail -4 > $O/tests.txt
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null; }
for i in 1 2; do
echo "c2 packed   $(b c2 20)" >> $O/tests.txt
echo "c2 lps64    $(GCPNET_SEGRED_LPS=64 b c2 20)" >> $O/tests.txt
done
echo "c3 packed   $(b c3 10)" >> $O/tests.txt
echo "c3 lps64    $(GCPNET_SEGRED_LPS=64 b c3 10)" >> $O/tests.txt
echo "c5 packed   $(b c5 4)" >> $O/tests.txt
echo "c5 lps64    $(GCPNET_SEGRED_LPS=64 b c5 4)" >> $O/tests.txt
cat $O/tests.txt
echo "--- sweep_layers seed 64 case 72 under different arithmetic" >> $O/tests.txt
for env in "X=1" "GCPNET_FUSE_AGG=0" "GCPNET_WG_FWD_B6=0" "GCPNET_CHAIN_FWD_FP32_MFMA=1" "GCPNET_TN_FP32=1" "GCPNET_SEGRED_LPS=64"; do
  echo "$env: $(env $env SWEEP_ONLY=72 timeout 600 python tests/sweep_layers.py 73 64 2>&1 | grep -v amdgpu | tail -2 | cut -c1-420 | tr '\n' ' ')" >> $O/tests.txt
done
tail -8 $O/tests.txt
