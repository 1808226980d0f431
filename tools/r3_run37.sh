set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_dconv_gpu.py tests/test_stages_gpu.py -q -m gpu -x -s > gpurun_out/r3/gpu_tests_37.log 2>&1 ; grep -a "unet bf16\]\|passed\|failed\|Error" gpurun_out/r3/gpu_tests_37.log | tail -8
timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_fp32.log 2>&1
HP_TIME_DCONV_PRECISION=bf16 timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_bf16.log 2>&1
paste -d'\n' gpurun_out/r3/dconv_layers_fp32.log gpurun_out/r3/dconv_layers_bf16.log | grep -v "amdgpu.ids\|calibration" | cut -c1-130
