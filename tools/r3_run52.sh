set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_xformers.py tests/test_sformer.py -q -m gpu -x > gpurun_out/r3/gpu_tests_52.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_52.log
