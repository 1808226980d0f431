mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_data_parallel_gpu.py tests/test_entry_points.py -q -m gpu -x > gpurun_out/r4/gpu_tests_5.log 2>&1; tail -3 gpurun_out/r4/gpu_tests_5.log
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra $EXTRA > gpurun_out/r4/bench_$name.json 2> gpurun_out/r4/bench_$name.err; python -c "
import json;d=json.load(open('gpurun_out/r4/bench_$name.json'));print('$name', d['ms_per_step'], d.get('unoverlapped_profiled_ms_per_step'))"; grep -h "peak HBM" gpurun_out/r4/bench_$name.err | tail -1; }
EXTRA="" run ws_on A=1
EXTRA="" run ws_on_hold2 HP_WGRAD_HOLD=2
EXTRA="" run ws_on_hold8 HP_WGRAD_HOLD=8
EXTRA="" run ws_on_mainhi HP_MAIN_PRIO=-1
