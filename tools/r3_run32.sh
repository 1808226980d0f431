set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_dconv_gpu.py tests/test_stages_gpu.py -q -m gpu -s -k "bf16" > gpurun_out/r3/gpu_tests_32.log 2>&1 ; grep -a "unet bf16\|passed\|failed\|Error\|^FAILED" gpurun_out/r3/gpu_tests_32.log | tail -30
