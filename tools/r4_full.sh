mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r4/gpu_tests_full.log 2>&1; tail -15 gpurun_out/r4/gpu_tests_full.log
timeout -k 10 900 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; tail -c 300 gpurun_out/r4/bench_default.json
