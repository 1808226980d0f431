mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_data_parallel_gpu.py tests/test_xformers.py tests/test_sformer.py tests/test_entry_points.py "tests/test_nlospose_gpu.py::test_train_step_benchmark_cube_512_batch2_vs_reference_golden" "tests/test_nlospose_gpu.py::test_bf16_storage_mode_vs_reference_golden_and_bf16_mode" -q -m gpu > gpurun_out/r4/gpu_tests_6.log 2>&1; tail -6 gpurun_out/r4/gpu_tests_6.log
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline $EXTRA > gpurun_out/r4/bench_$name.json 2> gpurun_out/r4/bench_$name.err; python -c "
import json;d=json.load(open('gpurun_out/r4/bench_$name.json'));print('$name', d['ms_per_step'], d.get('unoverlapped_profiled_ms_per_step'))"; }
EXTRA="--no-wgrad-stream" run d20_off A=1
EXTRA="" run d20_on A=1
EXTRA="--steps 10 --warmup 3" run d10_on A=1
EXTRA="--steps 40 --warmup 5" run d40_on A=1
