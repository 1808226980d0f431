set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -x > gpurun_out/r3/gpu_tests_gemm.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_gemm.log
timeout -k 10 500 python tools/time_conv_layers.py 512 128 4 fwd,dgrad fp32 l1.0,l1.1,l2.1,deconv1 > gpurun_out/r3/layers_gemm.log 2>&1; grep -v amdgpu gpurun_out/r3/layers_gemm.log | cut -c1-150
