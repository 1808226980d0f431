set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r3/gpu_tests_50.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_50.log
