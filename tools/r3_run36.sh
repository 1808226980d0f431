set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_nlospose_gpu.py tests/test_entry_points.py tests/test_highres_gpu.py -q -m gpu -s -k "bf16 or highres" > gpurun_out/r3/gpu_tests_36.log 2>&1 ; grep -a "bf16s\]\|unet bf16\]\|passed\|failed\|bf16s:\|fp32:\|Error" gpurun_out/r3/gpu_tests_36.log | tail -16
