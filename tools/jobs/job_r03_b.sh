cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_side_stream.py tests/test_trainer_glue.py tests/test_gpu_parity.py tests/test_full_size.py -m gpu -q -s 2>&1 | tail -80 > gpurun_out/r03_b/tests.txt
SWEEP_POISON=1 SWEEP_REPEAT=2 timeout 900 python -u tests/sweep_layers.py 80 1 2>&1 | tail -20 > gpurun_out/r03_b/sweep_layers_1.txt
timeout 300 python -u tools/repro/overread_v_in.py > gpurun_out/r03_b/repro_new.txt 2>&1; echo "rc $?" >> gpurun_out/r03_b/repro_new.txt
GCPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/repro/libgcpnet_hip_r02wgbwd.so timeout 300 python -u tools/repro/overread_v_in.py > gpurun_out/r03_b/repro_r02.txt 2>&1; echo "rc $?" >> gpurun_out/r03_b/repro_r02.txt
sleep 2
timeout 120 python -c "import torch; print('gpu alive', torch.ones(4, device='cuda').sum().item())" >> gpurun_out/r03_b/repro_r02.txt 2>&1
tail -n 5 gpurun_out/r03_b/repro_new.txt gpurun_out/r03_b/repro_r02.txt
