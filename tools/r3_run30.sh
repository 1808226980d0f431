set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_nlospose_gpu.py tests/test_entry_points.py tests/test_highres_gpu.py -q -m gpu -x -s > gpurun_out/r3/gpu_tests_30.log 2>&1 ; grep -a "bf16s\]\|passed\|failed\|bf16s:\|fp32:" gpurun_out/r3/gpu_tests_30.log | tail -12
