mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_headline_gpu.py tests/test_xformers.py tests/test_sformer.py tests/test_nlospose_gpu.py tests/test_entry_points.py -q -m gpu > gpurun_out/r4/gpu_tests_7.log 2>&1; tail -5 gpurun_out/r4/gpu_tests_7.log
HP_TIME_STATS=1 timeout -k 10 250 python tools/time_conv_layers.py 512 128 4 fwd bf16s l1,l2.0 > gpurun_out/r4/stats_cost_bf16s_slots.txt 2>&1; grep "l1.0.conv1\|l1.1.conv1\|l1.0.conv3\|l2.0.conv1" gpurun_out/r4/stats_cost_bf16s_slots.txt | cut -c1-150
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/r4/bench_$name.json 2> gpurun_out/r4/bench_$name.err; python -c "
import json;d=json.load(open('gpurun_out/r4/bench_$name.json'));print('$name', d['ms_per_step'], d.get('unoverlapped_profiled_ms_per_step'), d['hip_kernel_ms_per_step'].get('conv_igemm_k1'))"; }
run slots_fp32
run slots_bf16s --conv-precision bf16s
