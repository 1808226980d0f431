set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_dconv_gpu.py tests/test_stages_gpu.py -q -m gpu -x > gpurun_out/r3/gpu_tests_43.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_43.log
timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_fp32.log 2>&1
HP_STENCIL_ASYNC=0 timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_fp32_sync.log 2>&1
paste -d'\n' gpurun_out/r3/dconv_layers_fp32.log gpurun_out/r3/dconv_layers_fp32_sync.log | grep "1->1\|sum"
