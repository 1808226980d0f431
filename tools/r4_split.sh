mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_headline_gpu.py tests/test_entry_points.py -q -m gpu -x > gpurun_out/r4/gpu_tests_split.log 2>&1; tail -4 gpurun_out/r4/gpu_tests_split.log | cut -c1-200
for r0 in 0 d 0 d; do
  if [ $r0 = d ]; then unset HP_WGRAD_R0; else export HP_WGRAD_R0=$r0; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r4/bench_split_$r0.json 2> gpurun_out/r4/bench_split_$r0.err || exit 1
  python tools/show_bench.py gpurun_out/r4/bench_split_$r0.json | head -2
done
for r0 in 0 d; do
  if [ $r0 = d ]; then unset HP_WGRAD_R0; else export HP_WGRAD_R0=$r0; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --conv-precision bf16s > gpurun_out/r4/bench_split_bf16s_$r0.json 2> gpurun_out/r4/bench_split_bf16s_$r0.err || exit 1
  python tools/show_bench.py gpurun_out/r4/bench_split_bf16s_$r0.json | head -2
done
