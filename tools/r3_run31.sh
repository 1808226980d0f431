set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_stages_gpu.py -q -m gpu -x -s -k "unet_bf16" > gpurun_out/r3/gpu_tests_31.log 2>&1 ; grep -a "unet bf16\|passed\|failed\|Error\|assert" gpurun_out/r3/gpu_tests_31.log | tail -12
