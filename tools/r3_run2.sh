set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r3/gpu_tests_2.log 2>&1 ;
tail -5 gpurun_out/r3/gpu_tests_2.log
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_default_2.json 2> gpurun_out/r3/bench_default_2.err ;
tail -4 gpurun_out/r3/bench_default_2.err
