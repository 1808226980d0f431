set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_dconv_gpu.py tests/test_stages_gpu.py tests/test_highres_gpu.py -q -m gpu -x > gpurun_out/r3/gpu_tests_thin.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_thin.log
timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_fp32.log 2>&1
HP_TIME_DCONV_PRECISION=bf16 timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_bf16.log 2>&1
paste -d'\n' gpurun_out/r3/dconv_layers_fp32.log gpurun_out/r3/dconv_layers_bf16.log | grep -v "amdgpu.ids\|calibration" | cut -c1-130 | grep "sum\|4->4\|8->4 \|1->4\|8->8\|16->16"
