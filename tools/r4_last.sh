mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --deselect tests/test_data_parallel_gpu.py > gpurun_out/r4/gpu_tests_last3.log 2>&1; tail -4 gpurun_out/r4/gpu_tests_last3.log | cut -c1-220
for v in 1 2; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r4/bench_last.json 2> gpurun_out/r4/bench_last.err && python tools/show_bench.py gpurun_out/r4/bench_last.json | head -1 | cut -c1-50
done
timeout -k 10 300 python bench.py --workload highres --steps 10 --warmup 2 > gpurun_out/r4/bench_last_hr.json 2> gpurun_out/r4/bench_last_hr.err && python tools/show_bench.py gpurun_out/r4/bench_last_hr.json | head -1 | cut -c1-50
